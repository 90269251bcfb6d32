"""One total check of the 1M x 1M matcher against the CPU oracle: EVERY query of both directions of the
BASELINE configs[1] pair (seed 566, the pair's real FPFH rows from the HIP feature stages) through oracle.match_bf_subset --
the exhaustive matchBF restatement (include/matching.h:594-634: bf blocks of 200 000, later block wins a tie, lowest index inside
a block, NaN rows never match) -- against the tables of the production schedule (pruned, coarse-rejecting, re-filtered).

    python tools/full_match_check.py [--points 1000000] [--chunk 50000] [--out gpurun_out/full_match_check.json]

About 1e12 distance evaluations per direction: ~6 min on the 16 host cores of a one-GPU box.  Prints a progress line per chunk.
The one-line JSON result is what gets committed under profiles/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--chunk", type=int, default=50_000)
    ap.add_argument("--block", type=int, default=200_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_match_check.json"))
    ap.add_argument("--seed", type=int, default=None, help="generator seed (default: the bench pair's)")
    ap.add_argument("--scene", default="bench", choices=["bench", "planar"], help="bench: height field + boxes; planar: the planar-dominated scene of matcher_extremes")
    a = ap.parse_args()
    import torch
    import oracle as o
    from lgr_amd import capi, synthetic
    o.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    try:
        tok = open("/sys/fs/cgroup/cpu.max").read().split()
        if tok[0] != "max":
            cores = min(cores, max(1, round(int(tok[0]) / int(tok[1]))))
    except Exception:
        pass
    o.set_num_threads(cores)
    ctx = capi.Context(0)
    seed = synthetic.SEED if a.seed is None else a.seed
    pair = synthetic.make_planar_pair(a.points, seed=seed) if a.scene == "planar" else synthetic.make_pair(a.points, seed=seed)
    voxel = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
    feats = []
    for side in ("src", "tgt"):
        cloud = torch.from_numpy(pair[side]).cuda()
        nrm = ctx.normals_knn(ctx.downsample(cloud, voxel).clone(), 30, vp=pair["vp_" + side])
        feats.append(ctx.fpfh(cloud, nrm, 0.25))
    tabs = [x.cpu().numpy() for x in ctx.match_bf2(feats[0], feats[1], a.block)]
    ctx.sync()
    work, fmt, stats = ctx.match_work(), ctx.match_format(), ctx.match_stats()
    fh = [f.cpu().numpy() for f in feats]
    out = {"workload": "%s, seed %d, %d points per cloud" % ("BASELINE configs[1] pair" if a.scene == "bench" else "planar-dominated scene", seed, a.points), "bf_block_size": a.block,
           "irregular_rows": list(ctx.match_irregular()),
           "executed_tile_fraction": work, "operand_format": fmt, "dense_fallbacks": [int(stats["dense_ab"]), int(stats["dense_ba"])],
           "cores": cores, "directions": {}}
    t_all = time.time()
    total_bad = 0
    for name, q, t, gi, gd in (("src->tgt", fh[0], fh[1], tabs[0], tabs[1]), ("tgt->src", fh[1], fh[0], tabs[2], tabs[3])):
        n = q.shape[0]
        bad_i = bad_d = nomatch = 0
        t0 = time.time()
        for lo in range(0, n, a.chunk):
            sel = np.arange(lo, min(n, lo + a.chunk), dtype=np.int32)
            oi, od = o.match_bf_subset(q, sel, t, a.block)
            ok = oi >= 0
            bad_i += int((gi[sel] != oi).sum())
            bad_d += int((gd[sel].view(np.uint32)[ok] != od.view(np.uint32)[ok]).sum())
            nomatch += int((~ok).sum())
            print("%s: %d / %d queries, %d index + %d distance mismatches, %.0f s" % (name, sel[-1] + 1, n, bad_i, bad_d, time.time() - t0), flush=True)
        out["directions"][name] = {"queries": int(n), "train_rows": int(t.shape[0]), "index_mismatches": bad_i, "distance_bit_mismatches": bad_d,
                                   "queries_without_match": nomatch, "oracle_seconds": time.time() - t0}
        total_bad += bad_i + bad_d
    out["queries_total"] = int(fh[0].shape[0] + fh[1].shape[0])
    out["mismatches_total"] = total_bad
    out["oracle_seconds_total"] = time.time() - t_all
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(json.dumps(out) + "\n")
    print(json.dumps(out), flush=True)
    ctx.close()
    raise SystemExit(0 if total_bad == 0 else 4)


if __name__ == "__main__":
    main()
