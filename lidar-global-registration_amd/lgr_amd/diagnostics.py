"""Diagnostics shared by tests/test_gpu_concurrent_contexts.py and tools/exp_concurrent_stages.py: a bit-level snapshot of what one
lgr_align_dev call left in its context's workspace (surface clouds with their normals, FPFH rows, match tables, thresholds,
correspondences) + the result record, and a comparison that names the first stage whose output differs."""
import ctypes as C

import numpy as np

from . import capi

ORDER = ["ws_surf_s", "ws_surf_t", "ws_feat_s", "ws_feat_t", "ws_thr", "ws_ij", "ws_dij", "ws_ji", "ws_dji", "ws_corr", "align", "align_T"]
_hip = None


def _memcpy_dtoh(host, ptr, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    assert _hip.hipMemcpy(C.c_void_p(host.ctypes.data), ptr, C.c_size_t(nbytes), 2) == 0


def align_snapshot(ctx, src, tgt, params, surf_prefix=600000):
    """run ctx.align(src, tgt, params) and copy the pipeline buffers of that call to the host ({name: ndarray}, names in ORDER)"""
    full = ctx.align(src, tgt, params)
    n = int(src.shape[0])
    out = {}
    for name, count, dt in (("surf_s", 0, np.float32), ("surf_t", 0, np.float32), ("feat_s", n * 33, np.float32), ("feat_t", n * 33, np.float32),
                            ("thr", 2 * n, np.float32), ("ij", n, np.int32), ("dij", n, np.float32), ("ji", n, np.int32), ("dji", n, np.float32),
                            ("corr", full.n_correspondences * 4, np.int32)):
        ptr, cap = C.c_void_p(), C.c_size_t()
        assert capi.lib().lgr_debug_ws(ctx.h, name.encode(), C.byref(ptr), C.byref(cap)) == 0
        if count == 0:
            count = min(surf_prefix, n // 2) * 12          # the surface clouds: a prefix (their sizes are not exported)
        assert count * 4 <= cap.value, (name, count * 4, cap.value)
        host = np.empty(count, dt)
        if count:
            _memcpy_dtoh(host, ptr, count * 4)
        out["ws_" + name] = host
    out["align"] = np.array([full.iterations, full.n_inliers, full.n_correspondences], np.int64)
    out["align_T"] = full.matrix()
    return out


def first_difference(ref, got):
    """None when every buffer is bit-equal, else (name, differing bytes) of the first one in pipeline order"""
    for name in ORDER:
        x, y = ref[name], got[name]
        if x.shape != y.shape:
            return name, -1
        xb, yb = np.ascontiguousarray(x).reshape(-1).view(np.uint8), np.ascontiguousarray(y).reshape(-1).view(np.uint8)
        if not np.array_equal(xb, yb):
            return name, int((xb != yb).sum())
    return None


def gpu_serial():
    """serial number of GPU 0 as rocm-smi prints it ('' when unavailable): ties an observation to a physical unit"""
    import re
    import subprocess
    try:
        t = subprocess.run(["rocm-smi", "--showserial"], capture_output=True, text=True, timeout=30).stdout
        m = re.search(r"Serial Number:\s*(\S+)", t)
        return m.group(1) if m else ""
    except Exception:
        return ""


def box_identity():
    """what identifies the unit and software an observation was made on: GPU serial, VBIOS / firmware versions (rocm-smi), ROCm and HIP runtime
    versions -- printed by tests/test_gpu_concurrent_contexts.py on every run, so that a difference arrives with its box identified"""
    import subprocess
    out = {"gpu_serial": gpu_serial()}
    for key, argv in (("vbios", ["rocm-smi", "--showvbios"]), ("firmware", ["rocm-smi", "--showfwinfo"]), ("driver", ["rocm-smi", "--showdriverversion"])):
        try:
            t = subprocess.run(argv, capture_output=True, text=True, timeout=30).stdout
            out[key] = " | ".join(ln.strip() for ln in t.splitlines() if "GPU[0]" in ln or "Driver version" in ln)[:1500]
        except Exception:
            out[key] = ""
    try:
        out["rocm"] = open("/opt/rocm/.info/version").read().strip()
    except Exception:
        out["rocm"] = ""
    try:
        import torch
        out["torch_hip"] = str(torch.version.hip)
        out["device"] = torch.cuda.get_device_name(0)
    except Exception:
        pass
    return out
