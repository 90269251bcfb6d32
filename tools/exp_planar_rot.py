"""On-box experiment (round 5): what do the few all-zero FPFH rows of the planar scene cost the matcher?
python tools/exp_planar_rot.py [n_points]
Times lgr_match_bf2_dev on the planar pair's FPFH rows as they are, and with every row whose 11-bin blocks do not sum to 100 replaced by
NaN (= excluded from matching): the second figure is what an exact side lane for such rows could reach."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lidar-global-registration_amd")]
import numpy as np
import torch
from lgr_amd import capi, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ctx = capi.Context(0)
r = 0.25
voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))
pp = synthetic.make_planar_pair(n, seed=synthetic.SEED)
fs = []
for side in ("src", "tgt"):
    cloud = torch.from_numpy(pp[side]).cuda()
    surf = ctx.downsample(cloud, voxel).clone()
    fs.append(ctx.fpfh(cloud, ctx.normals_knn(surf.clone(), 30, vp=pp["vp_" + side]), r))
ctx.sync(); torch.cuda.synchronize()


def timed(a, b, tag):
    for _ in range(3):
        ctx.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.match_bf2(a, b, 200000)
        ctx.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{tag}: match stage {1e3 * dt:.2f} ms, mfma {ctx.match_kernel_ms():.2f} ms, tiles {ctx.match_work():.4f}, format {ctx.match_format()}, irregular {ctx.match_irregular()}", flush=True)


timed(fs[0], fs[1], "as is")
gs = []
for f in fs:
    bs = f.reshape(-1, 3, 11).sum(2)
    bad = ((bs - 100.0).abs() > 1e-2).any(1) & torch.isfinite(f).all(1)
    print("irregular rows:", int(bad.sum()), "of", f.shape[0], "; NaN rows:", int((~torch.isfinite(f).all(1)).sum()))
    g = f.clone(); g[bad] = float("nan")
    gs.append(g)
timed(gs[0], gs[1], "irregular rows excluded")
