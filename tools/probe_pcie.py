"""host-buffer (PCIe-inclusive) rate of lgr_align vs the resident-in-HBM rate; for DESIGN.md only."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi, synthetic
sys.path.insert(0, ROOT)
import bench
ctx = capi.Context(0)
pair = synthetic.make_pair(1000000)
p = bench.make_params(capi, pair, "lr")
src = torch.from_numpy(pair["src"]).cuda(); tgt = torch.from_numpy(pair["tgt"]).cuda()
for name, fn in (("resident", lambda: ctx.align(src, tgt, p)), ("host buffers", lambda: ctx.align_host(pair["src"], pair["tgt"], p))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); fn(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
    print(f"{name}: {dt*1e3:.1f} ms/pair  {1/dt:.3f} registrations/s")
