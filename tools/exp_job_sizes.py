"""On-box experiment (round 5): per-stage times of lgr_align_dev when the pair size changes from call to call (the configs[2] job shape).
python tools/exp_job_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-global-registration_amd")]
import numpy as np
import torch
from lgr_amd import capi, synthetic
import bench

ctx = capi.Context(0)
sizes = [980000, 970000, 950000, 940000, 930000, 890000, 880000, 1000000, 860000, 800000]
pairs = {}
for n in sorted(set(sizes)):
    pairs[n] = synthetic.make_pair(n, seed=566 + n)
for n in sizes:
    pair = pairs[n]
    src = torch.from_numpy(pair["src"]).cuda(); tgt = torch.from_numpy(pair["tgt"]).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ctx.align(src, tgt, bench.make_params(capi, pair, "lr"))
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    st = [round(float(x), 2) for x in res.stage_ms[:6]]
    print(f"n = {n:8d}: {dt:7.2f} ms  (time_cs + time_te {1e3 * (res.time_cs + res.time_te):7.2f})  stages [down, normals, fpfh, match, filter, ransac] = {st}", flush=True)
