import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, oracle
from lgr_amd import capi, synthetic
ctx = capi.Context(0)
src = synthetic.make_pair(20000, seed=7)["src"]
for voxel in (0.1, 5.0, 0.3):
    want = oracle.downsample(src, voxel)
    t = torch.from_numpy(src).cuda()
    outbuf = torch.full((src.shape[0], 12), -7.0, device="cuda")
    import ctypes as C
    n = C.c_int(0)
    ctx.check(capi.lib().lgr_downsample_dev(ctx.h, C.c_void_p(t.data_ptr()), src.shape[0], C.c_float(voxel), C.c_void_p(outbuf.data_ptr()), C.byref(n)))
    torch.cuda.synchronize()
    got = outbuf[:n.value].cpu().numpy()
    bad = np.where((want.view(np.uint32) != got.view(np.uint32)).any(1))[0]
    print(voxel, want.shape, got.shape, len(bad))
    for b in bad[:8]:
        print("  row", b, "want w", want[b, 8], want[b, :3], "got w", got[b, 8], got[b, :3])
