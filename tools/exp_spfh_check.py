"""Self-check of the SPFH bin filter (lgr_features.hip: pair_bins_fast2) on the 1M bench pair: a library built with -DLGR_SPFH_CHECK
evaluates BOTH the filter and the canonical pcl::computePairFeatures sequence for every (point, neighbour) pair and counts
  pairs | pairs the filter left undecided | decided pairs whose bins differ from the canonical ones (must be 0).

    bash tools/exp_spfh_check.sh            # builds build/var_spfhcheck/liblgr_hip.so (CPU container)
    LGR_HIP_LIB=build/var_spfhcheck/liblgr_hip.so python tools/exp_spfh_check.py [--points 1000000]      # GPU box
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    a = ap.parse_args()
    import torch
    from lgr_amd import capi, synthetic
    lib = capi.lib()
    assert hasattr(lib, "lgr_debug_spfh_check"), "needs a library built with -DLGR_SPFH_CHECK (tools/exp_spfh_check.sh)"
    ctx = capi.Context(0)
    out = {}
    voxel = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
    for seed in (synthetic.SEED, synthetic.SEED + 1):
        pair = synthetic.make_pair(a.points, seed=seed)
        for side in ("src", "tgt"):
            cloud = torch.from_numpy(pair[side]).cuda()
            nrm = ctx.normals_knn(ctx.downsample(cloud, voxel).clone(), 30, vp=pair["vp_" + side])
            cnt = (C.c_ulonglong * 4)()
            assert lib.lgr_debug_spfh_check(cnt, 1) == 0
            ctx.fpfh(cloud, nrm, 0.25)
            ctx.sync()
            assert lib.lgr_debug_spfh_check(cnt, 1) == 0
            out["seed%d_%s" % (seed, side)] = {"pairs": int(cnt[0]), "undecided": int(cnt[1]), "undecided_fraction": cnt[1] / max(1, cnt[0]), "decided_but_wrong": int(cnt[2])}
            print(side, out["seed%d_%s" % (seed, side)], flush=True)
    print(json.dumps(out))
    ctx.close()
    raise SystemExit(0 if all(v["decided_but_wrong"] == 0 for v in out.values()) else 5)


if __name__ == "__main__":
    main()
