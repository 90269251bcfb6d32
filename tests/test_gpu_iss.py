"""GPU parity: ISS key points (reference src/common.cpp:657-691 -> pcl::ISSKeypoint3D) and the
key-point path of the correspondence search / alignment vs the oracle.  Bar: indices bit-exact (integer work driven by
double-precision scatter matrices and Jacobi eigenvalues restated op for op), correspondences and transforms bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def pair():
    from lgr_amd import synthetic
    return synthetic.make_pair(30000, seed=21)


@pytest.mark.parametrize("radius", [0.05, 0.1, 0.2])
def test_iss_indices(lgr, oracle, pair, radius):
    for name in ("src", "tgt"):
        pts = pair[name]
        ref = oracle.iss_keypoints(pts, radius)
        got = lgr.iss_keypoints(cuda(pts), radius).cpu().numpy()
        assert np.array_equal(got, ref), (len(got), len(ref))
        assert len(ref) > 0 and np.all(np.diff(ref) > 0)


def test_iss_edge_cases(lgr, oracle):
    rng = np.random.default_rng(3)
    from lgr_amd import synthetic
    pts = synthetic.make_points(rng.uniform(-1, 1, (5000, 3)))
    pts[10, 0] = np.nan; pts[11, 2] = np.inf          # non-finite points: no neighbours, never key points
    pts[200] = pts[201]                                 # exact duplicates: equal saliency, both may survive the strict <
    for radius, mn in ((0.12, 4), (0.3, 4), (0.02, 4), (0.12, 40)):
        ref = oracle.iss_keypoints(pts, radius, min_neighbors=mn)
        got = lgr.iss_keypoints(cuda(pts), radius, min_neighbors=mn).cpu().numpy()
        assert np.array_equal(got, ref)
    assert len(lgr.iss_keypoints(cuda(pts[:1]), 0.1).cpu().numpy()) == 0
    import ctypes as C
    from lgr_amd import capi
    idx = np.zeros(5000, np.int32); n = C.c_int(0)
    rc = capi._lib.lgr_iss_keypoints(lgr.h, pts.ctypes.data_as(C.c_void_p), 5000, C.c_float(0.12), C.c_float(0.975), C.c_float(0.975), 4,
                                     idx.ctypes.data_as(C.c_void_p), C.byref(n))
    assert rc == 0 and np.array_equal(idx[: n.value], oracle.iss_keypoints(pts, 0.12))
    assert capi._lib.lgr_iss_keypoints(lgr.h, pts.ctypes.data_as(C.c_void_p), 5000, C.c_float(0.0), C.c_float(0.975), C.c_float(0.975), 4,
                                       idx.ctypes.data_as(C.c_void_p), C.byref(n)) != 0     # salient radius must be > 0 (iss_debug.cpp:98)


@pytest.mark.parametrize("matching", [0, 1, 2])
def test_correspondences_with_iss(lgr, oracle, pair, matching):
    from lgr_amd import capi
    kw = dict(matching_id=matching, bf_block_size=200000, distance_thr=0.1, keypoint_id=1, iss_radius_src=0.06, iss_radius_tgt=0.06,
              vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    ocorr, _ = oracle.correspondences(pair["src"], pair["tgt"], oracle.default_params(**kw))
    corr = lgr.correspondences(cuda(pair["src"]), cuda(pair["tgt"]), capi.default_params(**kw)).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    assert len(corr) == len(ocorr) and len(corr) > 20
    np.testing.assert_array_equal(corr["index_query"], ocorr["query"])
    np.testing.assert_array_equal(corr["index_match"], ocorr["match"])
    np.testing.assert_array_equal(bits(corr["distance"]), bits(ocorr["distance"]))
    np.testing.assert_array_equal(bits(corr["threshold"]), bits(ocorr["threshold"]))
    ks = oracle.iss_keypoints(pair["src"], 0.06)
    assert np.all(np.isin(corr["index_query"], ks))          # queries are key points of the source cloud


def test_align_with_iss(lgr, oracle, pair):
    from lgr_amd import capi
    kw = dict(matching_id=0, bf_block_size=200000, max_iterations=100000, distance_thr=0.1, keypoint_id=1, iss_radius_src=0.06,
              iss_radius_tgt=0.06, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    ores, ocorr, _ = oracle.align(pair["src"], pair["tgt"], oracle.default_params(rng_mode=oracle.RNG_PHILOX, **kw))
    res = lgr.align(cuda(pair["src"]), cuda(pair["tgt"]), capi.default_params(**kw))
    assert res.n_correspondences == len(ocorr)
    assert res.iterations == ores.iterations and res.n_inliers == ores.n_inliers and res.converged == ores.converged
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))
