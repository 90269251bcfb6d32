"""GPU parity: GROR initial alignment (BASELINE config 5; reference include/gror/ia_gror.hpp via alignGror,
src/alignment.cpp:21-35) vs the oracle.

Bars: node degrees, inlier masks and counts bit-exact (integers); the final 4x4 is a float sequence restated op for
op, compared bit-exact, and within 1e-2 / 1e-3 of the synthetic ground truth (GROR refines over noisy inliers).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_orc_corr(oracle, corr):
    out = np.zeros(corr.shape[0], oracle.CORR_DTYPE)
    out["query"] = corr["index_query"]; out["match"] = corr["index_match"]
    out["distance"] = corr["distance"]; out["threshold"] = corr["threshold"]
    return out


def problem(c, frac, seed, n_pts=20000, sigma=0.01):
    from lgr_amd import synthetic
    return synthetic.make_correspondence_problem(n_pts=n_pts, c=c, inlier_frac=frac, seed=seed, sigma=sigma)


@pytest.mark.parametrize("c,res", [(1, 0.05), (257, 0.05), (3000, 0.05), (7001, 0.2)])
def test_node_degree(lgr, oracle, c, res):
    pr = problem(c, 0.3, 11 + c)
    ctx = lgr
    deg = ctx.gror_node_degree(cuda(pr["src"]), cuda(pr["tgt"]), pr["corr"], res)
    ref = oracle.gror_node_degree(pr["src"], pr["tgt"], to_orc_corr(oracle, pr["corr"]), res)
    assert np.array_equal(deg, ref)
    if c > 1000:
        assert deg.max() > 0.2 * c      # the true correspondences vote for each other


@pytest.mark.parametrize("c,frac,seed", [(6000, 0.4, 3), (6000, 0.05, 4), (800, 0.5, 5), (799, 0.5, 6), (300, 0.3, 7), (20000, 0.02, 8)])
def test_gror_matches_oracle(lgr, oracle, c, frac, seed):
    pr = problem(c, frac, seed)
    ctx = lgr
    res, mask = ctx.gror(cuda(pr["src"]), cuda(pr["tgt"]), pr["corr"], 0.05)
    T_o, d = oracle.gror(pr["src"], pr["tgt"], to_orc_corr(oracle, pr["corr"]), 0.05, 800)
    assert res.estimated_iters == d["K"] == min(c, 800)
    assert int(res.metric) == d["best_count"]
    assert res.best_iteration == d["tcfs_rows"]
    assert res.n_inliers == d["n_inliers"] == int(mask.sum())
    assert np.array_equal(bits(res.matrix()), bits(T_o)), np.abs(res.matrix() - T_o).max()
    assert res.iterations == 1 and res.converged == 1
    # ground truth is recovered whenever a consistent set was found
    assert d["best_count"] > 10
    assert np.abs(res.matrix()[:3, :3] - pr["T_gt"][:3, :3]).max() < 1e-3
    assert np.abs(res.matrix()[:3, 3] - pr["T_gt"][:3, 3]).max() < 1e-2


def test_gror_no_consistent_set(lgr, oracle):
    """All-random correspondences: no row reaches best_count > 3 -> identity guess, refinement over whatever is near."""
    pr = problem(1500, 0.0, 21)
    ctx = lgr
    res, mask = ctx.gror(cuda(pr["src"]), cuda(pr["tgt"]), pr["corr"], 0.05)
    T_o, d = oracle.gror(pr["src"], pr["tgt"], to_orc_corr(oracle, pr["corr"]), 0.05, 800)
    assert int(res.metric) == d["best_count"] and res.n_inliers == d["n_inliers"]
    if d["n_inliers"] > 0:
        assert np.array_equal(bits(res.matrix()), bits(T_o))


def test_gror_host_entry_and_mask(lgr, oracle):
    import ctypes as C
    pr = problem(2500, 0.3, 31)
    ctx = lgr
    res_d, mask_d = ctx.gror(cuda(pr["src"]), cuda(pr["tgt"]), pr["corr"], 0.05)
    from lgr_amd import capi
    res = capi.Result()
    mask = np.zeros(2500, np.uint8)
    corr = np.ascontiguousarray(pr["corr"])
    rc = capi._lib.lgr_gror(ctx.h, pr["src"].ctypes.data_as(C.c_void_p), pr["src"].shape[0], pr["tgt"].ctypes.data_as(C.c_void_p), pr["tgt"].shape[0],
                           corr.ctypes.data_as(C.c_void_p), 2500, C.c_float(0.05), 800, C.byref(res), mask.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert np.array_equal(bits(res.matrix()), bits(res_d.matrix())) and np.array_equal(mask, mask_d)
    # the mask is the refinement's inlier set under the pre-refinement transform: true pairs dominate it
    assert mask.sum() > 0.25 * 2500


def test_align_gror_end_to_end(lgr, oracle):
    """alignPointClouds with alignment_id = gror (src/alignment.cpp:92-101): same correspondences as the RANSAC path,
    then GROR with resolution = distance_thr."""
    from lgr_amd import capi, synthetic
    pair = synthetic.make_pair(20000, seed=11)
    kw = dict(matching_id=0, bf_block_size=200000, distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    p = capi.default_params(alignment_id=1, **kw)
    ctx = lgr
    s, t = cuda(pair["src"]), cuda(pair["tgt"])
    res = ctx.align(s, t, p)
    corr = ctx.correspondences(s, t, p).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    p_o = oracle.default_params(**kw)
    corr_o, _ = oracle.correspondences(pair["src"], pair["tgt"], p_o)
    assert np.array_equal(to_orc_corr(oracle, corr), corr_o) and len(corr_o) > 200
    T_o, d = oracle.gror(pair["src"], pair["tgt"], corr_o, 0.1, 800)
    assert res.n_correspondences == corr_o.shape[0]
    assert res.n_inliers == d["n_inliers"] and int(res.metric) == d["best_count"]
    assert np.array_equal(bits(res.matrix()), bits(T_o))
    assert np.abs(res.matrix() - pair["T_gt"]).max() < 5e-2
    res_h = ctx.align_host(pair["src"], pair["tgt"], p)
    assert np.array_equal(bits(res_h.matrix()), bits(res.matrix()))
