"""ctypes binding of liblgr_hip.so (the C ABI declared in include/lgr.h).

The HIP extension is mandatory: importing this module raises if the shared library is missing; nothing here (or
anywhere in the product) falls back to a CPU path.
"""
import ctypes as C
import os

import numpy as np

try:   # torch ships its own HIP runtime: it must be loaded BEFORE liblgr_hip.so pulls in libamdhip64
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover
    _torch = None

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
# LGR_HIP_LIB: another build of the same sources (A/B experiments of tools/: e.g. a library built with `make EXP=-D...`); never a fallback
LIB_PATH = os.environ.get("LGR_HIP_LIB") or os.path.join(_CSRC, "liblgr_hip.so")

MATCH_LR, MATCH_ONE_SIDED, MATCH_CLUSTER = 0, 1, 2
METRIC_CORRESPONDENCES, METRIC_UNIFORMITY, METRIC_CLOSEST_PLANE, METRIC_COMBINATION = 0, 1, 2, 3
KEYPOINT_ANY, KEYPOINT_ISS = 0, 1
SCORE_CONSTANT, SCORE_MAE, SCORE_MSE, SCORE_EXP = 0, 1, 2, 3
ALIGN_RANSAC, ALIGN_GROR = 0, 1
ORDER_REFERENCE, ORDER_CANONICAL = 0, 1

CORR_DTYPE = np.dtype([("index_query", "<i4"), ("index_match", "<i4"), ("distance", "<f4"), ("threshold", "<f4")])


class Params(C.Structure):
    _fields_ = [
        ("feature_nr_points", C.c_int32), ("normal_nr_points", C.c_int32), ("edge_thr_coef", C.c_float),
        ("distance_thr", C.c_float), ("feature_radius", C.c_float), ("scale_factor", C.c_float),
        ("confidence", C.c_float), ("bf_block_size", C.c_int32), ("cluster_k", C.c_int32),
        ("randomness", C.c_int32), ("n_samples", C.c_int32),
        ("alignment_id", C.c_int32), ("matching_id", C.c_int32), ("metric_id", C.c_int32), ("score_id", C.c_int32),
        ("max_iterations", C.c_int32), ("normals_available", C.c_int32), ("fix_seed", C.c_int32),
        ("has_vp_src", C.c_int32), ("has_vp_tgt", C.c_int32), ("vp_src", C.c_float * 3), ("vp_tgt", C.c_float * 3),
        ("ransac_batch", C.c_int32), ("seed", C.c_uint64),
        ("keypoint_id", C.c_int32), ("iss_radius_src", C.c_float), ("iss_radius_tgt", C.c_float), ("use_bfmatcher", C.c_int32),
        ("has_guess", C.c_int32), ("match_search_radius", C.c_float), ("guess", C.c_float * 16),
    ]


class Result(C.Structure):
    _fields_ = [
        ("transformation", C.c_float * 16), ("iterations", C.c_int32), ("converged", C.c_int32),
        ("n_inliers", C.c_int32), ("metric", C.c_float), ("best_metric_before_refit", C.c_float),
        ("best_iteration", C.c_int32), ("num_rejections", C.c_int32), ("estimated_iters", C.c_int32),
        ("n_correspondences", C.c_int32), ("time_cs", C.c_double), ("time_te", C.c_double),
        ("stage_ms", C.c_float * 12),
    ]

    def matrix(self):
        return np.array(self.transformation, dtype=np.float32).reshape(4, 4).T.copy()


class MatchOptions(C.Structure):
    """lgr_match_options (include/lgr.h): how the matcher runs, never what it returns."""
    _fields_ = [("prune", C.c_int32), ("leaves", C.c_int32), ("near", C.c_int32), ("operand_format", C.c_int32), ("box_bounds", C.c_int32),
                ("column_stage", C.c_int32), ("coarse_rejection", C.c_int32), ("rerank_refilter", C.c_int32), ("pair_cap", C.c_int32),
                ("poison_tables", C.c_int32), ("self_check", C.c_int32), ("shell_bound", C.c_int32), ("split_sweep", C.c_int32), ("kept_cap", C.c_int32), ("auto_dense", C.c_int32),
                ("irregular_rows", C.c_int32)]


class CtxOptions(C.Structure):
    """lgr_ctx_options (include/lgr.h): how the context uses host threads and streams, never what it returns."""
    _fields_ = [("helper_contexts", C.c_int32), ("concurrent_contexts", C.c_int32), ("arithmetic", C.c_int32), ("pcl_neighbour_cap", C.c_int32),
                ("ransac_schedule", C.c_int32), ("reserved", C.c_int32 * 3)]


RANSAC_SCHEDULE_DEFAULT, RANSAC_SCHEDULE_CHAIN, RANSAC_SCHEDULE_RESIDENT = 0, 1, 2   # lgr_ctx_options.ransac_schedule
ARITH_FAST, ARITH_PCL = 0, 1   # lgr_ctx_options.arithmetic (include/lgr.h): the FPFH weighting as one fused chain in grid order / exactly as PCL writes it
LIBM_ACOSF, LIBM_ATANF, LIBM_ATAN2F, LIBM_SINF, LIBM_COSF = 0, 1, 2, 3, 4


FORMAT_AUTO, FORMAT_F32, FORMAT_F16, FORMAT_F16R = -1, 0, 1, 2


class LgrError(RuntimeError):
    pass


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {_CSRC}` (or __graft_entry__.build()). "
            "The MI355X HIP extension is mandatory; there is no CPU fallback.")
    return C.CDLL(LIB_PATH)


ABI_VERSION = 5      # LGR_VERSION of the include/lgr.h these structures mirror

_lib = load()
_lib.lgr_last_error.restype = C.c_char_p
_lib.lgr_last_error.argtypes = [C.c_void_p]
if _lib.lgr_version() != ABI_VERSION:     # a stale liblgr_hip.so would read lgr_params / call lgr_match_last_* with another layout
    raise ImportError(f"{LIB_PATH} has ABI revision {_lib.lgr_version()}, this binding needs {ABI_VERSION}: rebuild it (make -C {_CSRC})")


def lib():
    return _lib


def default_params(**kw):
    p = Params()
    _lib.lgr_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("vp_src", "vp_tgt"):
            setattr(p, k, (C.c_float * 3)(*[float(x) for x in v]))
            setattr(p, "has_" + k, 1)
        elif k == "guess":      # 4x4, row/col indexed normally -> column-major 16
            p.guess = (C.c_float * 16)(*np.asarray(v, np.float32).T.reshape(16).tolist())
            p.has_guess = 1
        else:
            assert hasattr(p, k), k
            setattr(p, k, v)
    return p


def _ptr(t):
    """device (torch tensor) or host (numpy) pointer as void*."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data_as(C.c_void_p)
    return C.c_void_p(t.data_ptr())


class Context:
    """One lgr_ctx bound to a torch device + (by default) torch's current stream on it.  stream=-1 (LGR_STREAM_OWN): the context
    creates a non-blocking stream of its own -- torch's stream does not wait for it, so device outputs of the *_dev wrappers below
    must be consumed after ctx.sync() (the wrappers that read results back through torch do that themselves: _join)."""

    def __init__(self, device=0, stream=None):
        import torch
        self.torch = torch
        self.device = int(device)
        self.foreign_stream = stream is not None and stream != torch.cuda.current_stream(self.device).cuda_stream
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        rc = _lib.lgr_ctx_create(self.device, C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise LgrError(f"lgr_ctx_create failed: {rc}")
        self.h = h

    def close(self):
        if self.h:
            _lib.lgr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise LgrError(f"rc={rc}: {_lib.lgr_last_error(self.h).decode()}")

    def sync(self):
        self.check(_lib.lgr_ctx_sync(self.h))

    def _join(self):
        """before torch reads a buffer the library wrote: nothing to do on torch's own stream, a stream sync otherwise"""
        if self.foreign_stream:
            self.sync()

    def set_match_options(self, **kw):
        """defaults + overrides (no arguments: back to the defaults); returns the options now in force"""
        o = MatchOptions()
        _lib.lgr_match_default_options(C.byref(o))
        for k, v in kw.items():
            assert hasattr(o, k), k
            setattr(o, k, int(v))
        self.check(_lib.lgr_ctx_set_match_options(self.h, C.byref(o)))
        return o

    def match_options(self):
        o = MatchOptions()
        self.check(_lib.lgr_ctx_get_match_options(self.h, C.byref(o)))
        return o

    def set_options(self, **kw):
        """lgr_ctx_options: defaults + overrides (helper_contexts=0: one host thread, one stream); returns the options now in force"""
        o = CtxOptions()
        _lib.lgr_ctx_default_options(C.byref(o))
        for k, v in kw.items():
            assert hasattr(o, k), k
            setattr(o, k, int(v))
        self.check(_lib.lgr_ctx_set_options(self.h, C.byref(o)))
        return o

    def host_threads(self):
        """1 + the helper host threads this context has started (at most 2)"""
        n = C.c_int(0)
        self.check(_lib.lgr_ctx_host_threads(self.h, C.byref(n)))
        return n.value

    def workspace_bytes(self):
        out = C.c_uint64(0)
        self.check(_lib.lgr_ctx_workspace_bytes(self.h, C.byref(out)))
        return int(out.value)

    def _dev(self):
        return self.torch.device("cuda", self.device)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self._dev())

    # ---- matching -------------------------------------------------------------------------------------------
    def match_bf(self, q, t, block=10000):
        """q, t: cuda float32 [m,33].  returns (idx int32 [mq], dist float32 [mq])."""
        torch = self.torch
        q = q.contiguous(); t = t.contiguous()
        idx = self.empty((q.shape[0],), torch.int32)
        dist = self.empty((q.shape[0],), torch.float32)
        self.check(_lib.lgr_match_bf_dev(self.h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], int(block), _ptr(idx), _ptr(dist)))
        return idx, dist

    def match_bf2(self, a, b, block=10000):
        torch = self.torch
        a = a.contiguous(); b = b.contiguous()
        ab_i = self.empty((a.shape[0],), torch.int32); ab_d = self.empty((a.shape[0],), torch.float32)
        ba_i = self.empty((b.shape[0],), torch.int32); ba_d = self.empty((b.shape[0],), torch.float32)
        self.check(_lib.lgr_match_bf2_dev(self.h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], int(block),
                                          _ptr(ab_i), _ptr(ab_d), _ptr(ba_i), _ptr(ba_d)))
        return ab_i, ab_d, ba_i, ba_d

    def sort_pairs_u32(self, keys, vals, begin_bit=0, end_bit=32):
        """lgr_sort_pairs_u32_dev: cuda int32 keys (bit patterns of the uint32 keys) and int32 values -> (keys, values) stably sorted
        by the key bits [begin_bit, end_bit)"""
        torch = self.torch
        keys = keys.contiguous(); vals = vals.contiguous()
        ko = self.empty((keys.shape[0],), torch.int32); vo = self.empty((keys.shape[0],), torch.int32)
        self.check(_lib.lgr_sort_pairs_u32_dev(self.h, _ptr(keys), _ptr(ko), _ptr(vals), _ptr(vo), C.c_size_t(keys.shape[0]), int(begin_bit), int(end_bit)))
        return ko, vo

    def sort_pairs_u64(self, keys, vals, ranges):
        """lgr_sort_pairs_u64_dev: cuda int64 keys (bit patterns), int32 values, ranges = [(shift, width), ...] least significant first"""
        torch = self.torch
        keys = keys.contiguous(); vals = vals.contiguous()
        ko = self.empty((keys.shape[0],), torch.int64); vo = self.empty((keys.shape[0],), torch.int32)
        sh = (C.c_int * max(1, len(ranges)))(*[r[0] for r in ranges]); wd = (C.c_int * max(1, len(ranges)))(*[r[1] for r in ranges])
        self.check(_lib.lgr_sort_pairs_u64_dev(self.h, _ptr(keys), _ptr(ko), _ptr(vals), _ptr(vo), C.c_size_t(keys.shape[0]), sh, wd, len(ranges)))
        return ko, vo

    def match_flann(self, q, t):
        """matchFLANN<FPFH> (include/matching.h:565-592): cuda float32 [m,33] -> (idx, dist)"""
        torch = self.torch
        q = q.contiguous(); t = t.contiguous()
        idx = self.empty((q.shape[0],), torch.int32)
        dist = self.empty((q.shape[0],), torch.float32)
        self.check(_lib.lgr_match_flann_dev(self.h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(idx), _ptr(dist)))
        return idx, dist

    def match_local(self, qpts, tpts, qf, tf, guess, radius):
        """matchLocal<FPFH> (include/matching.h:637-678): points [m,12], descriptors [m,33] on the device; guess 4x4 (numpy)"""
        torch = self.torch
        g = (C.c_float * 16)(*np.asarray(guess, np.float32).T.reshape(16).tolist())
        idx = self.empty((qpts.shape[0],), torch.int32)
        dist = self.empty((qpts.shape[0],), torch.float32)
        self.check(_lib.lgr_match_local_dev(self.h, _ptr(qpts), qpts.shape[0], _ptr(tpts), tpts.shape[0], _ptr(qf.contiguous()), _ptr(tf.contiguous()),
                                            g, C.c_float(radius), _ptr(idx), _ptr(dist)))
        return idx, dist

    def match_bf_host(self, q, t, block=10000):
        q = np.ascontiguousarray(q, np.float32); t = np.ascontiguousarray(t, np.float32)
        idx = np.zeros(q.shape[0], np.int32); dist = np.zeros(q.shape[0], np.float32)
        self.check(_lib.lgr_match_bf(self.h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], int(block), _ptr(idx), _ptr(dist)))
        return idx, dist

    def match_stats(self):
        out = (C.c_uint * 6)()
        self.check(_lib.lgr_match_last_stats(self.h, out))
        return dict(items_ab=out[0], dense_ab=out[1], items_ba=out[2], dense_ba=out[3], sub_cols=out[4], rg_rows=out[5])

    def match_work(self):
        """fraction of the (row block x column stage) tiles the MFMA passes of the last match call computed"""
        f = C.c_double(1.0)
        self.check(_lib.lgr_match_last_work(self.h, C.byref(f)))
        return f.value

    def selfcheck_rcp(self, lo, hi):
        """(differing, tested) over every float in [lo, hi]: the FPFH weighting kernel's reciprocal against the IEEE division"""
        out = (C.c_ulonglong * 2)()
        lo_b, hi_b = (int(np.float32(v).view(np.uint32)) for v in (lo, hi))
        self.check(_lib.lgr_selfcheck_rcp(self.h, C.c_uint(lo_b), C.c_uint(hi_b), out))
        return int(out[0]), int(out[1])

    def match_lbstats(self):
        """(zero, finite) lower bounds among the (row block, leaf) pairs of the last pruned match call: what lgr_match_options.auto_dense decides on"""
        out = (C.c_double * 2)()
        self.check(_lib.lgr_match_last_lbstats(self.h, out))
        return out[0], out[1]

    def match_irregular(self):
        """(query side, train side, gave up) of the last match call: rows that went through the exact side scan (lgr_match_options.irregular_rows)"""
        out = (C.c_uint * 3)()
        self.check(_lib.lgr_match_last_irregular(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def selfcheck_philox(self, key, counter4):
        """one Philox4x32-10 block from the device's generator (full counter; key = k0 | k1 << 32)"""
        c = (C.c_uint32 * 4)(*[int(x) for x in counter4])
        out = (C.c_uint32 * 4)()
        self.check(_lib.lgr_selfcheck_philox(self.h, C.c_uint64(key), c, out))
        return list(out)

    def selfcheck_libm(self, fn, a, b=None):
        """csrc/lgr_libm.cuh (glibc 2.35's float acosf / atanf / atan2f / sinf / cosf restated) evaluated on the device, element-wise"""
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(a if b is None else b, np.float32)
        out = np.empty_like(a)
        self.check(_lib.lgr_selfcheck_libm(self.h, int(fn), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_longlong(a.size), out.ctypes.data_as(C.c_void_p)))
        return out

    def match_issued(self):
        """the same with the stages of every pass summed (>= match_work): the MFMA work that was issued"""
        out = (C.c_double * 2)()
        self.check(_lib.lgr_match_last_issued(self.h, out))
        return out[0]

    def match_issued_pairs(self):
        """(row, column) element pairs of the padded operands in the stages the passes of the last match call issued"""
        out = (C.c_double * 2)()
        self.check(_lib.lgr_match_last_issued(self.h, out))
        return out[1]

    def match_pairs(self):
        """(query->train, train->query) pairs the MFMA re-filter of the rerank handed to the exact distance in the last match call"""
        out = (C.c_uint * 2)()
        self.check(_lib.lgr_match_last_pairs(self.h, out))
        return out[0], out[1]

    def match_coarse(self):
        """(tiles tested, tiles abandoned) by the coarse rejection inside the MFMA filter kernel in the last match call"""
        out = (C.c_double * 2)()
        self.check(_lib.lgr_match_last_coarse(self.h, out))
        return out[0], out[1]

    def match_shell(self):
        """tiles of the swept stages the shell test left out before any MFMA step in the last match call"""
        out = C.c_double()
        self.check(_lib.lgr_match_last_shell(self.h, C.byref(out)))
        return out.value

    def match_format(self):
        """'f16' (split operands on the f16 MFMA, K = 112), 'f16r' (the same on 30 rotated coordinates, K = 96) or 'f32'
        for the last match call"""
        v = C.c_int(0)
        self.check(_lib.lgr_match_last_format(self.h, C.byref(v)))
        return {0: "f32", 1: "f16", 2: "f16r"}[v.value]

    def match_check(self):
        """(rows, cols) worst |filtered - exact| / eps of the last match call run under set_match_options(self_check=1), or -1"""
        out = (C.c_double * 2)()
        self.check(_lib.lgr_match_last_check(self.h, out))
        return out[0], out[1]

    def match_kernel_ms(self):
        ms = C.c_float(0)
        self.check(_lib.lgr_match_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    # ---- geometry stages ------------------------------------------------------------------------------------
    def bbox(self, pts):
        out = self.empty((6,), self.torch.float32)
        self.check(_lib.lgr_bbox_dev(self.h, _ptr(pts), pts.shape[0], _ptr(out)))
        return out

    def knn(self, q, pts, k):
        torch = self.torch
        idx = self.empty((q.shape[0], k), torch.int32); d2 = self.empty((q.shape[0], k), torch.float32)
        self.check(_lib.lgr_knn_dev(self.h, _ptr(q), q.shape[0], _ptr(pts), pts.shape[0], int(k), _ptr(idx), _ptr(d2)))
        return idx, d2

    def smoothed_densities(self, pts, k=2):
        out = self.empty((pts.shape[0],), self.torch.float32)
        self.check(_lib.lgr_smoothed_densities_dev(self.h, _ptr(pts), pts.shape[0], int(k), _ptr(out)))
        return out

    def preprocess(self, pts, vp=None, normals_available=False):
        out = self.empty((pts.shape[0], 12), self.torch.float32)
        n = C.c_int(0)
        voxel = C.c_float(0)
        v = (C.c_float * 3)(*[float(x) for x in vp]) if vp is not None else None
        self.check(_lib.lgr_preprocess_dev(self.h, _ptr(pts), pts.shape[0], v, int(normals_available), _ptr(out), C.byref(n), C.byref(voxel)))
        return out[: n.value], voxel.value

    def preprocess_host(self, pts, vp=None, normals_available=False, order=ORDER_CANONICAL):
        pts = np.ascontiguousarray(pts, np.float32)
        out = np.zeros_like(pts)
        n = C.c_int(0)
        voxel = C.c_float(0)
        v = (C.c_float * 3)(*[float(x) for x in vp]) if vp is not None else None
        self.check(_lib.lgr_preprocess(self.h, _ptr(pts), pts.shape[0], v, int(normals_available), int(order), _ptr(out), C.byref(n), C.byref(voxel)))
        return out[: n.value].copy(), voxel.value

    def dedupe(self, pts):
        out = self.empty((max(pts.shape[0], 1), 12), self.torch.float32)
        n = C.c_int(0)
        self.check(_lib.lgr_dedupe_dev(self.h, _ptr(pts), pts.shape[0], _ptr(out), C.byref(n)))
        return out[: n.value]

    def cloud_density(self, pts, quantile=0.8):
        v = C.c_float(0)
        self.check(_lib.lgr_cloud_density_dev(self.h, _ptr(pts), pts.shape[0], C.c_float(quantile), C.byref(v)))
        return v.value

    def iss_keypoints(self, pts, radius, gamma21=0.975, gamma32=0.975, min_neighbors=4):
        idx = self.empty((max(pts.shape[0], 1),), self.torch.int32)
        n = C.c_int(0)
        self.check(_lib.lgr_iss_keypoints_dev(self.h, _ptr(pts), pts.shape[0], C.c_float(radius), C.c_float(gamma21), C.c_float(gamma32),
                                              int(min_neighbors), _ptr(idx), C.byref(n)))
        return idx[: n.value]

    def downsample(self, pts, voxel):
        out = self.empty((pts.shape[0], 12), self.torch.float32)
        n = C.c_int(0)
        self.check(_lib.lgr_downsample_dev(self.h, _ptr(pts), pts.shape[0], C.c_float(voxel), _ptr(out), C.byref(n)))
        return out[: n.value]

    def downsample_host(self, pts, voxel, order=ORDER_CANONICAL):
        pts = np.ascontiguousarray(pts, np.float32)
        out = np.zeros_like(pts)
        n = C.c_int(0)
        self.check(_lib.lgr_downsample(self.h, _ptr(pts), pts.shape[0], C.c_float(voxel), int(order), _ptr(out), C.byref(n)))
        return out[: n.value].copy()

    def normals_knn(self, pts, k=30, surf=None, vp=None):
        """in place on the cuda tensor pts"""
        v = (C.c_float * 3)(*[float(x) for x in vp]) if vp is not None else None
        self.check(_lib.lgr_normals_knn_dev(self.h, _ptr(pts), pts.shape[0], _ptr(surf), 0 if surf is None else surf.shape[0],
                                            int(k), v, 0))
        return pts

    def fpfh(self, kps, surf, radius):
        out = self.empty((kps.shape[0], 33), self.torch.float32)
        self.check(_lib.lgr_fpfh_dev(self.h, _ptr(kps), kps.shape[0], _ptr(surf), surf.shape[0], C.c_float(radius), _ptr(out)))
        return out

    def fpfh_host(self, kps, surf, radius):
        kps = np.ascontiguousarray(kps, np.float32); surf = np.ascontiguousarray(surf, np.float32)
        out = np.zeros((kps.shape[0], 33), np.float32)
        self.check(_lib.lgr_fpfh(self.h, _ptr(kps), kps.shape[0], _ptr(surf), surf.shape[0], C.c_float(radius), _ptr(out)))
        return out

    # ---- filters / correspondence search -------------------------------------------------------------------
    def filter(self, matching_id, src, tgt, ij, dij, ji, dji, distance_thr, cluster_k=40):
        out = self.empty((src.shape[0], 4), self.torch.int32)
        n = C.c_int(0)
        self.check(_lib.lgr_filter_dev(self.h, int(matching_id), _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0],
                                       _ptr(ij), _ptr(dij), _ptr(ji), _ptr(dji), C.c_float(distance_thr), int(cluster_k),
                                       _ptr(out), C.byref(n)))
        self._join()
        return out[: n.value].cpu().numpy().view(CORR_DTYPE).reshape(-1)

    def correspondences(self, src, tgt, params):
        out = self.empty((src.shape[0], 4), self.torch.int32)
        n = C.c_int(0)
        self.check(_lib.lgr_correspondences_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], C.byref(params),
                                                _ptr(out), C.byref(n)))
        return out[: n.value]

    def stage_ms(self):
        out = (C.c_float * 12)()
        _lib.lgr_ctx_stage_ms(self.h, out)
        return list(out)

    # ---- RANSAC ---------------------------------------------------------------------------------------------
    def _corr_dev(self, corr):
        if isinstance(corr, np.ndarray):
            return self.torch.from_numpy(np.ascontiguousarray(corr).view(np.int32).reshape(-1, 4)).to(self._dev())
        return corr

    def ransac_samples(self, seed, first, n, n_corr, n_samples=3):
        out = self.empty((n, n_samples), self.torch.int32)
        self.check(_lib.lgr_ransac_samples_n_dev(self.h, C.c_uint64(seed), int(first), int(n), int(n_corr), int(n_samples), _ptr(out)))
        return out

    def evaluate(self, src, tgt, corr, T, metric_id=METRIC_UNIFORMITY, score_id=SCORE_MSE):
        corr = self._corr_dev(corr)
        c = corr.shape[0]
        mask = self.empty((max(c, 1),), self.torch.uint8)
        T16 = (C.c_float * 16)(*np.asarray(T, np.float32).T.reshape(16).tolist())
        ni, rm, me = C.c_int(0), C.c_float(0), C.c_float(0)
        self.check(_lib.lgr_evaluate_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], _ptr(corr), c, T16,
                                         int(metric_id), int(score_id), _ptr(mask), C.byref(ni), C.byref(rm), C.byref(me)))
        self._join()
        return mask[:c].cpu().numpy(), ni.value, rm.value, me.value

    def choose_best_hypothesis(self, src, tgt, corr, tns):
        corr = self._corr_dev(corr)
        n = len(tns)
        buf = np.ascontiguousarray(np.stack([np.asarray(T, np.float32).T.reshape(16) for T in tns]), np.float32) if n else np.zeros((1, 16), np.float32)
        out = (C.c_float * 16)()
        bi = C.c_int(-1)
        uni = np.zeros(max(n, 1), np.float32)
        self.check(_lib.lgr_choose_best_hypothesis_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], _ptr(corr), corr.shape[0],
                                                       _ptr(buf), n, out, C.byref(bi), _ptr(uni)))
        return bi.value, np.array(out, np.float32).reshape(4, 4).T.copy(), uni[:n].copy()

    def evaluate_plane(self, src, tgt, T, score_id=SCORE_CONSTANT, seed=566, counter=0, with_pairs=False):
        T16 = (C.c_float * 16)(*np.asarray(T, np.float32).T.reshape(16).tolist())
        n, rm, me, th, npairs = C.c_int(0), C.c_float(0), C.c_float(0), C.c_float(0), C.c_int(0)
        pairs = np.zeros((max(int(0.01 * src.shape[0]), 1), 2), np.int32) if with_pairs else None
        self.check(_lib.lgr_evaluate_plane_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], T16, int(score_id), C.c_uint64(seed),
                                               C.c_uint32(counter), C.byref(n), C.byref(rm), C.byref(me), C.byref(th), _ptr(pairs), C.byref(npairs)))
        out = dict(n_inl=n.value, rmse=rm.value, metric=me.value, thr=th.value)
        if with_pairs:
            out["pairs"] = pairs[: npairs.value].copy()
        return out

    def ransac_replay(self, src, tgt, corr, params, triples):
        torch = self.torch
        corr = self._corr_dev(corr)
        n = triples.shape[0]
        assert triples.shape[1] == params.n_samples, "one row of params.n_samples correspondence indices per hypothesis"
        ok = self.empty((n,), torch.uint8); Ts = self.empty((n, 16), torch.float32)
        ninl = self.empty((n,), torch.int32); met = self.empty((n,), torch.float32)
        self.check(_lib.lgr_ransac_replay_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], _ptr(corr), corr.shape[0],
                                              C.byref(params), _ptr(triples), n, _ptr(ok), _ptr(Ts), _ptr(ninl), _ptr(met)))
        self._join()
        return ok.cpu().numpy(), Ts.cpu().numpy(), ninl.cpu().numpy(), met.cpu().numpy()

    def ransac(self, src, tgt, corr, params):
        corr = self._corr_dev(corr)
        c = corr.shape[0]
        res = Result()
        mask = self.empty((max(c, 1),), self.torch.uint8)
        self.check(_lib.lgr_ransac_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], _ptr(corr), c,
                                       C.byref(params), C.byref(res), _ptr(mask)))
        self._join()
        return res, mask[:c].cpu().numpy()

    def gror(self, src, tgt, corr, resolution, k_optimal=800):
        corr = self._corr_dev(corr)
        c = corr.shape[0]
        res = Result()
        mask = self.empty((max(c, 1),), self.torch.uint8)
        self.check(_lib.lgr_gror_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], _ptr(corr), c,
                                     C.c_float(resolution), int(k_optimal), C.byref(res), _ptr(mask)))
        self._join()
        return res, mask[:c].cpu().numpy()

    def gror_node_degree(self, src, tgt, corr, resolution):
        corr = self._corr_dev(corr)
        c = corr.shape[0]
        deg = self.empty((max(c, 1),), self.torch.int32)
        self.check(_lib.lgr_gror_node_degree_dev(self.h, _ptr(src), _ptr(tgt), _ptr(corr), c, C.c_float(resolution), _ptr(deg)))
        self._join()
        return deg[:c].cpu().numpy()

    def refit(self, src, tgt, corr, mask=None):
        corr = self._corr_dev(corr)
        T = (C.c_float * 16)()
        self.check(_lib.lgr_refit_svd_dev(self.h, _ptr(src), _ptr(tgt), _ptr(corr), corr.shape[0], _ptr(mask), T))
        return np.array(T, np.float32).reshape(4, 4).T.copy()

    def align(self, src, tgt, params):
        res = Result()
        self.check(_lib.lgr_align_dev(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], C.byref(params), C.byref(res)))
        return res

    def align_host(self, src, tgt, params):
        src = np.ascontiguousarray(src, np.float32); tgt = np.ascontiguousarray(tgt, np.float32)
        res = Result()
        self.check(_lib.lgr_align(self.h, _ptr(src), src.shape[0], _ptr(tgt), tgt.shape[0], C.byref(params), C.byref(res)))
        return res
