"""On-box experiment (round 5): does a small kernel on another stream get onto the GPU while spfh_tile_kernel / fpfh_mfma_kernel run?
python tools/exp_corun.py -- times a chain of 20 tiny torch kernels (x += 1 on 64 K floats: 256-thread workgroups, no LDS, few registers) on a
second stream, alone and while lgr_fpfh_dev of a 1M-point cloud runs on the context's stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lidar-global-registration_amd")]
import numpy as np
import torch
from lgr_amd import capi, synthetic

ctx = capi.Context(0)      # torch's current stream
pair = synthetic.make_pair(1_000_000, seed=566)
voxel = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
cloud = torch.from_numpy(pair["src"]).cuda()
nrm = ctx.normals_knn(ctx.downsample(cloud, voxel).clone(), 30, vp=pair["vp_src"])
side = torch.cuda.Stream()
x = torch.zeros(65536, device="cuda")
y = torch.zeros(8 << 20, device="cuda")


z = torch.rand(1 << 20, device="cuda")


def chain(n=20, big=False):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        for _ in range(n):
            if big == "sort":
                z.sort()
            elif big == "cumsum":
                z.cumsum(0)
            else:
                (y if big else x).add_(1.0)
        e1.record()
    return e0, e1


for big in (False, True, "cumsum", "sort"):
    for trial in range(3):
        torch.cuda.synchronize()
        e0, e1 = chain(big=big)
        torch.cuda.synchronize()
        alone = e0.elapsed_time(e1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = ctx.fpfh(cloud, nrm, 0.25)          # enqueues the FPFH stage (sorts, spfh_tile_kernel, fpfh_mfma_kernel) on the main stream
        time.sleep(0.0008)                       # let the big kernels start
        e0, e1 = chain(big=big)
        ctx.sync(); torch.cuda.synchronize()
        print(f"{big if isinstance(big, str) else ('8M-element add' if big else '64K-element add')} chain of 20: alone {alone:.3f} ms, beside the FPFH stage {e0.elapsed_time(e1):.3f} ms (FPFH stage wall {1e3 * (time.perf_counter() - t0):.2f} ms)", flush=True)
