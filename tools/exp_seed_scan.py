"""On-box experiment (round 5): how much does the matcher's time depend on the scene?  Synthetic pairs of one size, several seeds:
python tools/exp_seed_scan.py N seed [seed ...]  -> per seed the matcher stage alone with its work statistics"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-global-registration_amd")]
import gc
import numpy as np
import torch
from lgr_amd import capi, synthetic

n = int(sys.argv[1]); seeds = [int(x) for x in sys.argv[2:]]
ctx = capi.Context(0)
if os.environ.get("NOGC"):
    gc.disable()
if os.environ.get("MATCH_OPTS"):   # e.g. MATCH_OPTS="near=48,shell_bound=0"
    ctx.set_match_options(**{k: int(v) for k, v in (kv.split("=") for kv in os.environ["MATCH_OPTS"].split(","))})
r = 0.25
voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))
for seed in seeds:
    pair = synthetic.make_pair(n, seed=seed)
    fs = []
    for side in ("src", "tgt"):
        cloud = torch.from_numpy(pair[side]).cuda()
        surf = ctx.downsample(cloud, voxel).clone()
        fs.append(ctx.fpfh(cloud, ctx.normals_knn(surf.clone(), 30, vp=pair["vp_" + side]), r))
    ctx.sync(); torch.cuda.synchronize()
    dts = []
    for _ in range(int(os.environ.get("REPS", "2"))):
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.match_bf2(fs[0], fs[1], 200000)
        ctx.sync(); torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        dts.append(round(dt, 2))
    print("  all runs:", dts)
    st = ctx.match_stats()
    print(f"seed {seed}: match stage {dt:.2f} ms, mfma {ctx.match_kernel_ms():.2f} ms, tiles {ctx.match_work():.4f} (issued {ctx.match_issued():.4f}), coarse (tested, abandoned) "
          f"{ctx.match_coarse()}, shell skipped {ctx.match_shell():.3g}, rerank [items_ab, dense_ab, items_ba, dense_ba, groups, rg_rows] {st}, pairs {ctx.match_pairs()}, "
          f"lb (zero, finite) {ctx.match_lbstats()}, NaN rows {int((~torch.isfinite(fs[0]).all(1)).sum())} + {int((~torch.isfinite(fs[1]).all(1)).sum())}, irregular {ctx.match_irregular()}", flush=True)
