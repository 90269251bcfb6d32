"""GPU: degenerate inputs and error behaviour of the boundary entry points (lgr_align*, lgr_correspondences*), through
the C ABI, against the oracle where a result exists.

The reference has no error codes: it prints and carries on (SURVEY 8b "Errors"); clouds too small to give three
correspondences come back as the identity, not converged (src/sac_prerejective_omp.cpp:36-42).  The ABI returns a
negative status only for arguments the reference could not even have been called with.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_VOXEL_TOO_SMALL = -1, -5, -6   # include/lgr.h


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def pair():
    from lgr_amd import synthetic
    return synthetic.make_pair(6000, seed=21)


def base_params(mod, pair, **kw):
    return mod.default_params(matching_id=0, bf_block_size=2000, max_iterations=20000, distance_thr=0.1,
                              vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"], **kw)


def test_tiny_clouds_are_identity_not_converged(lgr, oracle, pair):
    from lgr_amd import capi
    for n in (1, 2, 5):
        src, tgt = pair["src"][:n].copy(), pair["tgt"][:n].copy()
        res = lgr.align(cuda(src), cuda(tgt), base_params(capi, pair))
        assert res.converged == 0 and np.array_equal(res.matrix(), np.eye(4, dtype=np.float32))
        if n >= 2:
            ores, ocorr, _ = oracle.align(src, tgt, base_params(oracle, pair, rng_mode=oracle.RNG_PHILOX))
            assert ores.converged == 0 and np.array_equal(res.matrix(), ores.matrix())
            assert res.n_correspondences == len(ocorr)
        resh = lgr.align_host(src, tgt, base_params(capi, pair))
        assert resh.converged == 0 and np.array_equal(resh.matrix(), np.eye(4, dtype=np.float32))


def test_empty_cloud(lgr, pair):
    import torch
    from lgr_amd import capi
    empty = torch.zeros((0, 12), dtype=torch.float32, device="cuda")
    res = lgr.align(empty, cuda(pair["tgt"]), base_params(capi, pair))
    assert res.converged == 0 and res.n_correspondences == 0 and np.array_equal(res.matrix(), np.eye(4, dtype=np.float32))
    res = lgr.align(cuda(pair["src"]), empty, base_params(capi, pair))
    assert res.converged == 0 and res.n_correspondences == 0
    corr = lgr.correspondences(empty, empty, base_params(capi, pair))
    assert corr.shape[0] == 0


def test_nan_points_are_ignored_like_the_reference(lgr, oracle, pair):
    """NaN points do not move the bounding box (its std::min / std::max keep the other operand, include/common.h:266-280),
    are dropped by the voxel grid (isValid, src/downsample.cpp:25) and never get a feature; the correspondences of the
    remaining points equal the oracle's."""
    from lgr_amd import capi
    src, tgt = pair["src"].copy(), pair["tgt"].copy()
    src[17, 0] = np.nan; src[400, 2] = np.nan; tgt[5, 1] = np.nan; tgt[3000, :3] = np.nan
    corr = lgr.correspondences(cuda(src), cuda(tgt), base_params(capi, pair)).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    ocorr, _ = oracle.correspondences(src, tgt, base_params(oracle, pair))
    assert len(corr) == len(ocorr) > 50
    assert np.array_equal(corr["index_query"], ocorr["query"]) and np.array_equal(corr["index_match"], ocorr["match"])
    assert not np.isin([17, 400], corr["index_query"]).any() and not np.isin([5, 3000], corr["index_match"]).any()


def test_infinite_coordinate_is_a_clean_status(lgr, pair):
    """An infinite coordinate makes the reference's bounding box infinite and its voxel index (int) floor(inf) -- undefined
    behaviour there.  Here: a status code, no crash, and the context stays usable."""
    from lgr_amd import capi
    lib = capi.lib()
    src, tgt = pair["src"].copy(), pair["tgt"].copy()
    src[400, 2] = np.inf
    s, t = cuda(src), cuda(tgt)
    res = capi.Result()
    p = base_params(capi, pair)
    rc = lib.lgr_align_dev(lgr.h, C.c_void_p(s.data_ptr()), s.shape[0], C.c_void_p(t.data_ptr()), t.shape[0], C.byref(p), C.byref(res))
    assert rc in (0, ERR_VOXEL_TOO_SMALL)
    ok = lgr.align(cuda(pair["src"]), t, p)
    assert ok.n_correspondences > 0


def test_identical_clouds_give_the_identity(lgr, oracle, pair):
    from lgr_amd import capi
    src = pair["src"]
    p = base_params(capi, pair); p.vp_tgt[:] = p.vp_src[:]
    po = base_params(oracle, pair, rng_mode=oracle.RNG_PHILOX); po.vp_tgt[:] = po.vp_src[:]
    res = lgr.align(cuda(src), cuda(src.copy()), p)
    ores, ocorr, _ = oracle.align(src, src.copy(), po)
    assert res.converged == ores.converged == 1
    assert res.n_correspondences == len(ocorr) and res.n_inliers == ores.n_inliers
    assert np.array_equal(res.matrix().view(np.uint32), ores.matrix().view(np.uint32))
    assert np.abs(res.matrix() - np.eye(4)).max() < 1e-4


def test_invalid_arguments_return_status_not_crash(lgr, pair):
    from lgr_amd import capi
    lib = capi.lib()
    src, tgt = cuda(pair["src"]), cuda(pair["tgt"])
    p = base_params(capi, pair)
    res = capi.Result()
    ps, pt = C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr())
    assert lib.lgr_align_dev(None, ps, 10, pt, 10, C.byref(p), C.byref(res)) == ERR_INVALID_ARG          # no context
    assert lib.lgr_align_dev(lgr.h, ps, -1, pt, 10, C.byref(p), C.byref(res)) == ERR_INVALID_ARG        # negative size
    assert lib.lgr_align_dev(lgr.h, None, 10, pt, 10, C.byref(p), C.byref(res)) == ERR_INVALID_ARG      # null cloud
    assert lib.lgr_align_dev(lgr.h, ps, 10, pt, 10, None, C.byref(res)) == ERR_INVALID_ARG              # null params
    assert lib.lgr_align_dev(lgr.h, ps, 10, pt, 10, C.byref(p), None) == ERR_INVALID_ARG                # null result
    assert b"" != lib.lgr_last_error(lgr.h)
    bad = base_params(capi, pair); bad.n_samples = 9                                                          # kernels are instantiated for 3..8 samples
    assert lib.lgr_align_dev(lgr.h, ps, src.shape[0], pt, tgt.shape[0], C.byref(bad), C.byref(res)) == ERR_UNSUPPORTED
    bad = base_params(capi, pair); bad.alignment_id = 2                                                        # "teaser": alignTeaser throws in the reference
    assert lib.lgr_align_dev(lgr.h, ps, src.shape[0], pt, tgt.shape[0], C.byref(bad), C.byref(res)) == ERR_UNSUPPORTED
    # the pipeline entry points build the filter tables themselves -- cluster_k outside what the cluster filter keeps
    # (1 .. 64 spatial neighbours) and unknown matching ids are refused BEFORE any stage runs, not after a full pipeline run
    import time
    for field, value in (("cluster_k", 65), ("cluster_k", 0), ("cluster_k", -3), ("matching_id", 7)):
        bad = base_params(capi, pair); bad.matching_id = capi.MATCH_CLUSTER
        setattr(bad, field, value)
        n = C.c_int(-1)
        out = lgr.empty((src.shape[0], 4), src.dtype)
        t0 = time.perf_counter()
        assert lib.lgr_align_dev(lgr.h, ps, src.shape[0], pt, tgt.shape[0], C.byref(bad), C.byref(res)) == ERR_INVALID_ARG, (field, value)
        assert lib.lgr_correspondences_dev(lgr.h, ps, src.shape[0], pt, tgt.shape[0], C.byref(bad), C.c_void_p(out.data_ptr()), C.byref(n)) == ERR_INVALID_ARG
        assert time.perf_counter() - t0 < 0.05          # no stage ran
    okc = base_params(capi, pair); okc.matching_id = capi.MATCH_LR; okc.cluster_k = 65     # cluster_k is not read by the other filters (the reference ignores it too)
    assert lib.lgr_align_dev(lgr.h, ps, src.shape[0], pt, tgt.shape[0], C.byref(okc), C.byref(res)) == 0
    # the context stays usable after errors
    ok = lgr.align(src, tgt, p)
    assert ok.n_correspondences > 0


def test_out_of_range_correspondence_indices_are_an_error_not_a_fault(lgr, pair):
    """caller-supplied correspondences (alignRansac / alignGror / estimateOptimalRigidTransformation / the metric estimators take
    them from the caller, src/alignment.cpp:14-35): a stale index must come back as LGR_ERR_INVALID_ARG before any kernel
    gathers through it, and the context must stay usable."""
    from lgr_amd import capi, synthetic
    pr = synthetic.make_correspondence_problem(n_pts=5000, c=600, inlier_frac=0.5, seed=4)
    src, tgt = cuda(pr["src"]), cuda(pr["tgt"])
    good = pr["corr"]
    p = capi.default_params(max_iterations=2000, distance_thr=0.05)
    for field, value in (("index_query", 5000), ("index_match", 5000), ("index_query", -1), ("index_match", 2**31 - 1)):
        bad = good.copy()
        bad[field][317] = value
        for call in (lambda: lgr.ransac(src, tgt, bad, p), lambda: lgr.gror(src, tgt, bad, 0.05),
                     lambda: lgr.evaluate(src, tgt, bad, pr["T_gt"]), lambda: lgr.choose_best_hypothesis(src, tgt, bad, [pr["T_gt"]])):
            with pytest.raises(capi.LgrError) as e:
                call()
            assert f"rc={ERR_INVALID_ARG}" in str(e.value)
        T = (C.c_float * 16)()
        hb = np.ascontiguousarray(bad)
        rc = capi.lib().lgr_refit_svd(lgr.h, pr["src"].ctypes.data_as(C.c_void_p), pr["tgt"].ctypes.data_as(C.c_void_p), 5000, 5000,
                                      hb.ctypes.data_as(C.c_void_p), len(hb), T)
        assert rc == ERR_INVALID_ARG
    res, mask = lgr.ransac(src, tgt, good, p)            # the context is fine afterwards, and the good set registers
    assert res.converged == 1 and np.abs(res.matrix() - pr["T_gt"]).max() < 5e-2
    # the largest valid indices are accepted
    edge = good.copy()
    edge["index_query"][0] = 4999; edge["index_match"][0] = 4999
    lgr.ransac(src, tgt, edge, p)
