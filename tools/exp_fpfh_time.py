"""Wall time of the FPFH stage alone (lgr_fpfh_dev: key sorts, spfh_tile_kernel, fpfh_mfma_kernel) on the bench pair's source cloud,
best and median of N runs -- for A/B runs of variant libraries (LGR_HIP_LIB=...).
    python tools/exp_fpfh_time.py [--points 1000000] [--runs 9]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--runs", type=int, default=9)
    a = ap.parse_args()
    import torch
    from lgr_amd import capi, synthetic
    ctx = capi.Context(0)
    pair = synthetic.make_pair(a.points, seed=synthetic.SEED)
    voxel = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
    cloud = torch.from_numpy(pair["src"]).cuda()
    nrm = ctx.normals_knn(ctx.downsample(cloud, voxel).clone(), 30, vp=pair["vp_src"])
    ts = []
    for _ in range(a.runs + 1):
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = ctx.fpfh(cloud, nrm, 0.25)
        ctx.sync(); torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts = sorted(ts[1:])
    print("fpfh stage alone (%s): best %.3f ms, median %.3f ms over %d runs; checksum %.6f" % (os.environ.get("LGR_HIP_LIB", "in-tree"), ts[0], ts[len(ts) // 2], a.runs,
                                                                                              float(f.nan_to_num().double().sum().item())))
    ctx.close()


if __name__ == "__main__":
    main()
