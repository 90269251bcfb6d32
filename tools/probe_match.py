"""GPU probe: matcher timing + candidate statistics on FPFH-like rows (not a test, not the bench)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi

def fpfh_like(rng, m):
    x = rng.gamma(0.6, 1.0, (m, 3, 11)) + 1e-3
    x = 100.0 * x / x.sum(2, keepdims=True)
    return x.reshape(m, 33).astype(np.float32)

ctx = capi.Context(0)
rng = np.random.default_rng(0)
for m in [int(a) for a in sys.argv[1:]] or [100000, 400000]:
    a = torch.from_numpy(fpfh_like(rng, m)).cuda(); b = torch.from_numpy(fpfh_like(rng, m)).cuda()
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        r = ctx.match_bf2(a, b, 200000); ctx.sync()
        dt = time.time() - t
        print(f"m={m} it={it} bf2 {dt*1e3:.1f} ms  {69.0*m*m/dt/1e12:.1f} TFLOP/s(alg)  stats={ctx.match_stats()} kernel_ms={ctx.match_kernel_ms():.1f}", flush=True)
