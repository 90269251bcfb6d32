"""ctypes binding of the CPU ORACLE (oracle/_build/liblgr_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblgr_oracle.so")

ORDER_LIBSTDCXX, ORDER_CANONICAL = 0, 1
METRIC_CORRESPONDENCES, METRIC_UNIFORMITY, METRIC_CLOSEST_PLANE, METRIC_COMBINATION = 0, 1, 2, 3
SCORE_CONSTANT, SCORE_MAE, SCORE_MSE, SCORE_EXP = 0, 1, 2, 3
MATCH_LR, MATCH_ONE_SIDED, MATCH_CLUSTER = 0, 1, 2
RNG_MT19937_LEMIRE, RNG_MT19937_REJECT, RNG_PHILOX = 0, 1, 2
KEYPOINT_ANY, KEYPOINT_ISS = 0, 1

CORR_DTYPE = np.dtype([("query", "<i4"), ("match", "<i4"), ("distance", "<f4"), ("threshold", "<f4")])


class Params(C.Structure):
    _fields_ = [
        ("feature_nr_points", C.c_int), ("normal_nr_points", C.c_int), ("edge_thr_coef", C.c_float),
        ("distance_thr", C.c_float), ("feature_radius", C.c_float), ("scale_factor", C.c_float),
        ("confidence", C.c_float), ("bf_block_size", C.c_int), ("cluster_k", C.c_int), ("n_samples", C.c_int),
        ("matching_id", C.c_int), ("metric_id", C.c_int), ("score_id", C.c_int), ("max_iterations", C.c_int),
        ("normals_available", C.c_int), ("has_vp_src", C.c_int), ("has_vp_tgt", C.c_int),
        ("vp_src", C.c_float * 3), ("vp_tgt", C.c_float * 3),
        ("keypoint_id", C.c_int), ("iss_radius_src", C.c_float), ("iss_radius_tgt", C.c_float),
        ("rng_mode", C.c_int), ("n_threads", C.c_int), ("batch_size", C.c_int), ("seed", C.c_uint64),
        ("use_bfmatcher", C.c_int), ("has_guess", C.c_int), ("match_search_radius", C.c_float), ("guess", C.c_float * 16),
    ]


class Result(C.Structure):
    _fields_ = [
        ("T", C.c_float * 16), ("iterations", C.c_int), ("converged", C.c_int), ("n_inliers", C.c_int),
        ("metric", C.c_float), ("best_metric_before_refit", C.c_float), ("best_iteration", C.c_int),
        ("num_rejections", C.c_int), ("estimated_iters", C.c_int),
    ]

    def matrix(self):
        """4x4 numpy matrix (row/col indexed normally)."""
        return np.array(self.T, dtype=np.float32).reshape(4, 4).T.copy()


def build(force=False):
    stale = not os.path.exists(_SO)
    if not stale and os.path.isdir(os.path.join(_HERE, "src")):
        so_t = os.path.getmtime(_SO)
        for r, _, fs in os.walk(_HERE):
            for f in fs:
                if f.endswith((".cpp", ".h")) and os.path.getmtime(os.path.join(r, f)) > so_t:
                    stale = True
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_atan2f.restype = C.c_float
        _lib.orc_atan2f.argtypes = [C.c_float, C.c_float]
        for n in ("orc_logf", "orc_cbrtf", "orc_expf"):
            getattr(_lib, n).restype = C.c_float
            getattr(_lib, n).argtypes = [C.c_float]
    return _lib


_REF_SO = os.path.join(_HERE, "_ref", "liblgr_ref_utils.so")
_ref = None


def ref_utils(build_if_possible=True):
    """The reference's own std-only translation units (src/utils.cpp, src/csv_parser.cpp + headers) behind the extern "C"
    shim oracle/ref/ref_utils_shim.cpp -> oracle/_ref/liblgr_ref_utils.so (recipe: `make -C oracle ref`; needs
    /root/reference, so it is built in the authoring container and travels to the GPU box as a built file).
    Returns the ctypes library or None when it is neither present nor buildable."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF_SO) and build_if_possible and os.path.isdir("/root/reference/include"):
            subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
        if not os.path.exists(_REF_SO):
            return None
        _ref = C.CDLL(_REF_SO)
        _ref.ref_combine_hash_int.restype = C.c_uint64
        _ref.ref_combine_hash_int.argtypes = [C.c_uint64, C.c_int]
        _ref.ref_combine_hash_float.restype = C.c_uint64
        _ref.ref_combine_hash_float.argtypes = [C.c_uint64, C.c_float]
        for n in ("ref_quantile_float",):
            getattr(_ref, n).restype = C.c_float
            getattr(_ref, n).argtypes = [C.c_double, C.c_void_p, C.c_int]
        for n in ("ref_mean_float", "ref_stddev_float"):
            getattr(_ref, n).restype = C.c_float
            getattr(_ref, n).argtypes = [C.c_void_p, C.c_int]
        _ref.ref_rng_stream.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
    return _ref


def comb_or_max(n, k):
    return int(lib().orc_comb_or_max(int(n), int(k)))


def voxel_hash(ix, iy, iz):
    f = lib().orc_voxel_hash
    f.restype = C.c_uint64
    return int(f(int(ix), int(iy), int(iz)))


def point_hash(x, y, z):
    f = lib().orc_point_hash
    f.restype = C.c_uint64
    f.argtypes = [C.c_float, C.c_float, C.c_float]
    return int(f(x, y, z))


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 12, a.shape
    return a


def make_points(xyz, normals=None, intensity=1.0):
    """n x 12 float32 array in pcl::PointXYZINormal layout."""
    xyz = np.asarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    p = np.zeros((n, 12), dtype=np.float32)
    p[:, 0:3] = xyz
    p[:, 3] = 1.0
    if normals is not None:
        p[:, 4:7] = normals
    p[:, 8] = intensity
    return p


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("vp_src", "vp_tgt"):
            setattr(p, k, (C.c_float * 3)(*v))
            setattr(p, "has_" + k, 1)
        elif k == "guess":      # 4x4, row/col indexed normally -> column-major 16
            p.guess = (C.c_float * 16)(*np.asarray(v, np.float32).T.reshape(16).tolist())
            p.has_guess = 1
        else:
            assert hasattr(p, k), k
            setattr(p, k, v)
    return p


# bit mask (lgr_oracle.h): single pieces of PCL 1.12.1's own arithmetic can be switched alone
ARITH_ROUND4, ARITH_PCL_EIGEN33, ARITH_PCL_ACOS, ARITH_PCL_W_ORDER, ARITH_PCL_W_ROUND, ARITH_PCL_W_NORM, ARITH_PCL_ATAN2 = 0, 1, 2, 4, 8, 16, 32
ARITH_PCL_LIBM, ARITH_PCL_WEIGHTING, ARITH_PCL = 2 | 32, 4 | 8 | 16, 63
ARITH_CANONICAL = 1 | 2 | 32   # the default since round 5: PCL's own normals and pair features; the weighting in grid order (what the HIP default restates)
ARITH_PIECES = {"eigen33_normals": 1, "acosf_swap_test": 2, "weighting_neighbour_order": 4, "weighting_rounded_product": 8,
                "weighting_running_normaliser": 16, "atan2f_f1": 32}
LIBM_ACOSF, LIBM_ATANF, LIBM_ATAN2F, LIBM_SINF, LIBM_COSF = 0, 1, 2, 3, 4


def set_arith_mode(mode):
    """ARITH_CANONICAL (default: what the HIP library's default mode restates -- PCL's eigen33 normals and glibc 2.35 pair features, the
    FPFH weighting as one fused chain in grid order), ARITH_PCL (also PCL 1.12.1's own weighting order and rounding steps: what
    lgr_ctx_options.arithmetic = LGR_ARITH_PCL restates), ARITH_ROUND4 (rounds 1-4's canonical orders), or any mask of pieces"""
    lib().orc_set_arith_mode(int(mode))


def libm_eval(fn, a, b=None):
    """orc_libm.h (glibc 2.35's acosf / atanf / atan2f / sinf / cosf restated) element-wise"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.empty_like(a)
    lib().orc_libm_eval(int(fn), _p(a), _p(b), C.c_long(a.size), _p(out))
    return out


def libm_check_range(fn, lo_bits, hi_bits):
    """number of floats with bits in [lo_bits, hi_bits] whose restated result differs from the running libm's"""
    f = lib().orc_libm_check_range
    f.restype = C.c_long
    return int(f(int(fn), C.c_uint32(lo_bits), C.c_uint32(hi_bits)))


def libm_check_atan2(y, x):
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    f = lib().orc_libm_check_atan2
    f.restype = C.c_long
    return int(f(_p(y), _p(x), C.c_long(y.size)))


def arith_mode():
    return int(lib().orc_arith_mode())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


def bbox(pts):
    pts = _pts(pts)
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    lib().orc_bbox(_p(pts), pts.shape[0], _p(mn), _p(mx))
    return mn, mx


def downsample(pts, voxel, order=ORDER_CANONICAL):
    pts = _pts(pts)
    out = np.zeros_like(pts)
    n = C.c_int(0)
    rc = lib().orc_downsample(_p(pts), pts.shape[0], C.c_float(voxel), order, _p(out), C.byref(n))
    assert rc == 0, rc
    return out[: n.value].copy()


def normals_knn(pts, k=30, surf=None, vp=None, normals_available=False):
    pts = _pts(pts).copy()
    s = _pts(surf) if surf is not None else None
    v = np.ascontiguousarray(vp, dtype=np.float32) if vp is not None else None
    rc = lib().orc_normals_knn(_p(pts), pts.shape[0], _p(s), 0 if s is None else s.shape[0], k, _p(v), int(normals_available))
    assert rc == 0
    return pts


def fpfh(kps, surf, radius, libm=False):
    kps, surf = _pts(kps), _pts(surf)
    out = np.zeros((kps.shape[0], 33), np.float32)
    rc = lib().orc_fpfh(_p(kps), kps.shape[0], _p(surf), surf.shape[0], C.c_float(radius), _p(out), int(libm))
    assert rc == 0
    return out


def spfh(surf, radius, libm=False):
    surf = _pts(surf)
    out = np.zeros((surf.shape[0], 33), np.float32)
    rc = lib().orc_spfh(_p(surf), surf.shape[0], C.c_float(radius), _p(out), int(libm))
    assert rc == 0
    return out


def match_bf(q, t, block=10000):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    idx = np.zeros(q.shape[0], np.int32)
    dist = np.zeros(q.shape[0], np.float32)
    rc = lib().orc_match_bf(_p(q), q.shape[0], _p(t), t.shape[0], block, _p(idx), _p(dist))
    assert rc == 0
    return idx, dist


def match_bf_subset(q, qsel, t, block=10000):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    qsel = np.ascontiguousarray(qsel, np.int32)
    idx = np.zeros(qsel.shape[0], np.int32)
    dist = np.zeros(qsel.shape[0], np.float32)
    rc = lib().orc_match_bf_subset(_p(q), _p(qsel), qsel.shape[0], _p(t), t.shape[0], block, _p(idx), _p(dist))
    assert rc == 0
    return idx, dist


def match_flann(q, t):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    idx = np.zeros(q.shape[0], np.int32)
    dist = np.zeros(q.shape[0], np.float32)
    assert lib().orc_match_flann(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist)) == 0
    return idx, dist


def match_local(qpts, tpts, qf, tf, guess, radius):
    qpts, tpts = _pts(qpts), _pts(tpts)
    qf = np.ascontiguousarray(qf, np.float32)
    tf = np.ascontiguousarray(tf, np.float32)
    g = (C.c_float * 16)(*np.asarray(guess, np.float32).T.reshape(16).tolist())
    idx = np.zeros(qpts.shape[0], np.int32)
    dist = np.zeros(qpts.shape[0], np.float32)
    assert lib().orc_match_local(_p(qpts), qpts.shape[0], _p(tpts), tpts.shape[0], _p(qf), _p(tf), g, C.c_float(radius), _p(idx), _p(dist)) == 0
    return idx, dist


def inverse4(T):
    a = (C.c_float * 16)(*np.asarray(T, np.float32).T.reshape(16).tolist())
    out = (C.c_float * 16)()
    lib().orc_inverse4(a, out)
    return np.array(out, np.float32).reshape(4, 4).T.copy()


def knn(qpts, pts, k):
    qpts, pts = _pts(qpts), _pts(pts)
    idx = np.zeros((qpts.shape[0], k), np.int32)
    d2 = np.zeros((qpts.shape[0], k), np.float32)
    rc = lib().orc_knn(_p(qpts), qpts.shape[0], _p(pts), pts.shape[0], k, _p(idx), _p(d2))
    assert rc == 0
    return idx, d2


def smoothed_densities(pts, k=2):
    pts = _pts(pts)
    out = np.zeros(pts.shape[0], np.float32)
    rc = lib().orc_smoothed_densities(_p(pts), pts.shape[0], k, _p(out))
    assert rc == 0, rc
    return out


def cloud_density(pts, quantile=0.8):
    pts = _pts(pts)
    out = C.c_float(0)
    rc = lib().orc_cloud_density(_p(pts), pts.shape[0], C.c_float(quantile), C.byref(out))
    assert rc == 0
    return out.value


def filter_matches(matching_id, src, tgt, ij_idx, ij_dist, ji_idx, ji_dist, distance_thr, cluster_k=40):
    src, tgt = _pts(src), _pts(tgt)
    ij_idx = np.ascontiguousarray(ij_idx, np.int32)
    ji_idx = np.ascontiguousarray(ji_idx, np.int32)
    ij_dist = np.ascontiguousarray(ij_dist, np.float32)
    ji_dist = np.ascontiguousarray(ji_dist, np.float32)
    out = np.zeros(src.shape[0], CORR_DTYPE)
    n = C.c_int(0)
    rc = lib().orc_filter(matching_id, _p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(ij_idx), _p(ij_dist),
                          _p(ji_idx), _p(ji_dist), C.c_float(distance_thr), cluster_k, _p(out), C.byref(n))
    assert rc == 0, rc
    return out[: n.value].copy()


def correspondences(src, tgt, params):
    src, tgt = _pts(src), _pts(tgt)
    out = np.zeros(src.shape[0], CORR_DTYPE)
    n = C.c_int(0)
    st = np.zeros(8, np.float64)
    rc = lib().orc_correspondences(_p(src), src.shape[0], _p(tgt), tgt.shape[0], C.byref(params), _p(out), C.byref(n), _p(st))
    assert rc == 0, rc
    return out[: n.value].copy(), st


def rng_stream(mode, seed, n):
    out = np.zeros(n, np.int32)
    lib().orc_rng_stream(mode, C.c_uint64(seed), n, _p(out))
    return out


def philox(seed, it):
    out = (C.c_uint32 * 4)()
    lib().orc_philox(C.c_uint64(seed), C.c_uint32(it), out)
    return list(out)


def philox_full(key, counter4):
    """Philox4x32-10 with the full 128-bit counter; key = k0 | k1 << 32"""
    c = (C.c_uint32 * 4)(*[int(x) for x in counter4])
    out = (C.c_uint32 * 4)()
    lib().orc_philox_full(C.c_uint64(key), c, out)
    return list(out)


def select3(r, n_corr):
    rr = (C.c_int * 3)(*[int(x) for x in r])
    s = (C.c_int * 3)()
    lib().orc_select3(rr, int(n_corr), s)
    return list(s)


def select_n(r, n_corr):
    n = len(r)
    rr = (C.c_int * n)(*[int(x) for x in r])
    s = (C.c_int * n)()
    lib().orc_select_n(rr, n, int(n_corr), s)
    return list(s)


def philox_draws(seed, it, n_samples):
    r = (C.c_int * n_samples)()
    lib().orc_philox_draws(C.c_uint64(seed), C.c_uint32(it), int(n_samples), r)
    return list(r)


def poly_ok_n(src, tgt, sidx, tidx, edge_thr=0.95):
    src, tgt = _pts(src), _pts(tgt)
    n = len(sidx)
    return bool(lib().orc_poly_ok_n(_p(src), _p(tgt), (C.c_int * n)(*sidx), (C.c_int * n)(*tidx), n, C.c_float(edge_thr)))


def umeyama_n(src, tgt, sidx, tidx):
    src, tgt = _pts(src), _pts(tgt)
    n = len(sidx)
    T = np.zeros(16, np.float32)
    lib().orc_umeyama_n(_p(src), _p(tgt), (C.c_int * n)(*sidx), (C.c_int * n)(*tidx), n, _p(T))
    return T.reshape(4, 4).T.copy()


def poly_ok(src, tgt, sidx, tidx, edge_thr=0.95):
    src, tgt = _pts(src), _pts(tgt)
    return bool(lib().orc_poly_ok(_p(src), _p(tgt), (C.c_int * 3)(*sidx), (C.c_int * 3)(*tidx), C.c_float(edge_thr)))


def umeyama3(src, tgt, sidx, tidx):
    src, tgt = _pts(src), _pts(tgt)
    T = np.zeros(16, np.float32)
    lib().orc_umeyama3(_p(src), _p(tgt), (C.c_int * 3)(*sidx), (C.c_int * 3)(*tidx), _p(T))
    return T.reshape(4, 4).T.copy()


def _T16(T):
    """4x4 (normal indexing) -> 16 floats column-major."""
    return np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(16))


def evaluate(src, tgt, corr, T, metric_id=METRIC_UNIFORMITY, score_id=SCORE_MSE):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    mask = np.zeros(corr.shape[0], np.uint8)
    n_inl, rmse, metric = C.c_int(0), C.c_float(0), C.c_float(0)
    t16 = _T16(T)
    lib().orc_evaluate(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(corr), corr.shape[0], _p(t16), metric_id,
                       score_id, _p(mask), C.byref(n_inl), C.byref(rmse), C.byref(metric))
    return mask, n_inl.value, rmse.value, metric.value


def estimate_max_iterations(src, tgt, corr, T, confidence=0.999, nr_samples=3):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    t16 = _T16(T)
    return lib().orc_estimate_max_iterations(_p(src), _p(tgt), _p(corr), corr.shape[0], _p(t16), C.c_float(confidence), nr_samples)


def refit(src, tgt, corr, mask):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    mask = np.ascontiguousarray(mask, np.uint8)
    T = np.zeros(16, np.float32)
    lib().orc_refit(_p(src), _p(tgt), _p(corr), corr.shape[0], _p(mask), _p(T))
    return T.reshape(4, 4).T.copy()


def replay(src, tgt, corr, params, triples):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    triples = np.ascontiguousarray(triples, np.int32)
    n = triples.shape[0]
    ok = np.zeros(n, np.uint8)
    Ts = np.zeros((n, 16), np.float32)
    ninl = np.zeros(n, np.int32)
    met = np.zeros(n, np.float32)
    lib().orc_replay(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(corr), corr.shape[0], C.byref(params),
                     _p(triples), n, _p(ok), _p(Ts), _p(ninl), _p(met))
    return ok, Ts, ninl, met


def ransac(src, tgt, corr, params):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    res = Result()
    mask = np.zeros(corr.shape[0], np.uint8)
    rc = lib().orc_ransac(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(corr), corr.shape[0], C.byref(params),
                          C.byref(res), _p(mask))
    assert rc == 0, rc
    return res, mask


def align(src, tgt, params):
    src, tgt = _pts(src), _pts(tgt)
    res = Result()
    corr = np.zeros(src.shape[0], CORR_DTYPE)
    n = C.c_int(0)
    st = np.zeros(8, np.float64)
    rc = lib().orc_align(_p(src), src.shape[0], _p(tgt), tgt.shape[0], C.byref(params), C.byref(res), _p(corr),
                         C.byref(n), _p(st))
    assert rc == 0, rc
    return res, corr[: n.value].copy(), st


def knnresult_run(capacity, dists, indices):
    dists = np.ascontiguousarray(dists, np.float32)
    indices = np.ascontiguousarray(indices, np.int32)
    oi = np.zeros(capacity, np.int32)
    od = np.zeros(capacity, np.float32)
    c = lib().orc_knnresult_run(capacity, _p(dists), _p(indices), dists.shape[0], _p(oi), _p(od))
    return oi[:c].copy(), od[:c].copy()


def update_hypotheses(tns, metrics, new_T, new_metric, distance_thr, cap=64):
    buf = np.zeros((cap, 16), np.float32)
    mb = np.zeros(cap, np.float32)
    n = len(metrics)
    for i in range(n):
        buf[i] = _T16(tns[i])
        mb[i] = metrics[i]
    m = lib().orc_update_hypotheses(_p(buf), _p(mb), n, cap, _p(_T16(new_T)), C.c_float(new_metric), C.c_float(distance_thr))
    assert m >= 0
    return [buf[i].reshape(4, 4).T.copy() for i in range(m)], [float(mb[i]) for i in range(m)]


def rot_trans_diff(T1, T2):
    a, t = C.c_float(0), C.c_float(0)
    lib().orc_rot_trans_diff(_p(_T16(T1)), _p(_T16(T2)), C.byref(a), C.byref(t))
    return a.value, t.value


def svd3(A):
    A = np.ascontiguousarray(A, np.float32).reshape(9)
    U, S, V = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().orc_svd3(_p(A), _p(U), _p(S), _p(V))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def atan2f(y, x):
    return lib().orc_atan2f(C.c_float(y), C.c_float(x))


def logf(x):
    return lib().orc_logf(C.c_float(x))


def cbrtf(x):
    return lib().orc_cbrtf(C.c_float(x))


def expf(x):
    return lib().orc_expf(C.c_float(x))


def gror_node_degree(src, tgt, corr, resolution):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    deg = np.zeros(corr.shape[0], np.int32)
    lib().orc_gror_node_degree(_p(src), _p(tgt), _p(corr), corr.shape[0], C.c_float(resolution), _p(deg))
    return deg


def gror(src, tgt, corr, resolution, K=800):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    T = np.zeros(16, np.float32)
    diag = np.zeros(8, np.int32)
    ang = C.c_float(0)
    rc = lib().orc_gror(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(corr), corr.shape[0], C.c_float(resolution), int(K),
                        _p(T), _p(diag), C.byref(ang))
    assert rc == 0
    return T.reshape(4, 4).T.copy(), dict(K=int(diag[0]), best_count=int(diag[1]), tcfs_rows=int(diag[2]), n_inliers=int(diag[3]), best_angle=ang.value)


def iss_keypoints(pts, radius, gamma21=0.975, gamma32=0.975, min_neighbors=4, with_third=False):
    pts = _pts(pts)
    idx = np.zeros(pts.shape[0], np.int32)
    n = C.c_int(0)
    third = np.zeros(pts.shape[0], np.float64) if with_third else None
    rc = lib().orc_iss_keypoints(_p(pts), pts.shape[0], C.c_float(radius), C.c_float(gamma21), C.c_float(gamma32), int(min_neighbors),
                                 _p(idx), C.byref(n), _p(third))
    assert rc == 0, rc
    return (idx[: n.value].copy(), third) if with_third else idx[: n.value].copy()


def eigvals3d(S6):
    S6 = np.ascontiguousarray(S6, np.float64)
    ev = np.zeros(3, np.float64)
    lib().orc_eigvals3d(_p(S6), _p(ev))
    return ev


def dedupe(pts, order=ORDER_CANONICAL):
    pts = _pts(pts)
    out = np.zeros_like(pts)
    n = C.c_int(0)
    assert lib().orc_dedupe(_p(pts), pts.shape[0], order, _p(out), C.byref(n)) == 0
    return out[: n.value].copy()


def preprocess(pts, vp=None, normals_available=False, order=ORDER_CANONICAL):
    pts = _pts(pts)
    out = np.zeros_like(pts)
    n = C.c_int(0)
    voxel = C.c_float(0)
    v = np.ascontiguousarray(vp, dtype=np.float32) if vp is not None else None
    rc = lib().orc_preprocess(_p(pts), pts.shape[0], _p(v), int(normals_available), order, _p(out), C.byref(n), C.byref(voxel))
    assert rc == 0, rc
    return out[: n.value].copy(), voxel.value


def evaluate_plane(src, tgt, T, score_id=SCORE_CONSTANT, seed=566, counter=0, with_pairs=False):
    src, tgt = _pts(src), _pts(tgt)
    T16 = np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(16))
    n, sc, rm, me, th = C.c_int(0), C.c_float(0), C.c_float(0), C.c_float(0), C.c_float(0)
    pairs = np.zeros((max(int(0.01 * src.shape[0]), 1), 2), np.int32) if with_pairs else None
    rc = lib().orc_evaluate_plane(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(T16), int(score_id), C.c_uint64(seed), C.c_uint32(counter),
                                  C.byref(n), C.byref(sc), C.byref(rm), C.byref(me), C.byref(th), _p(pairs))
    assert rc == 0, rc
    out = dict(n_inl=n.value, score=sc.value, rmse=rm.value, metric=me.value, thr=th.value)
    if with_pairs:
        out["pairs"] = pairs[: n.value].copy()
    return out


def choose_best_hypothesis(src, tgt, corr, tns):
    src, tgt = _pts(src), _pts(tgt)
    corr = np.ascontiguousarray(corr, CORR_DTYPE)
    buf = np.ascontiguousarray(np.stack([_T16(T) for T in tns]), np.float32) if len(tns) else np.zeros((1, 16), np.float32)
    out = np.zeros(16, np.float32)
    uni = np.zeros(max(len(tns), 1), np.float32)
    i = lib().orc_choose_best_hypothesis(_p(src), src.shape[0], _p(tgt), tgt.shape[0], _p(corr), corr.shape[0], _p(buf), len(tns), _p(out), _p(uni))
    return i, out.reshape(4, 4).T.copy(), uni[: len(tns)].copy()
