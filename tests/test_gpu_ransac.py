"""GPU parity: RANSAC pieces (sampler, prerejection, 3-point transform, inlier masks, metrics, refit, whole loop),
match filters and the end-to-end alignPointClouds path vs the oracle.

Bars: sample triples / ok flags / inlier counts / masks / correspondences bit-exact; transforms and metrics are
float sequences restated op for op -> compared bit-exact too (tolerance 0), with the north-star tolerance 1e-4 on
the final 4x4 asserted against the ground truth of the synthetic pair.
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


@pytest.fixture(scope="module")
def problem():
    from lgr_amd import synthetic
    return synthetic.make_correspondence_problem(n_pts=20000, c=6000, inlier_frac=0.4, seed=3)


def params_pair(oracle, capi, **kw):
    ok = dict(kw)
    p_o = oracle.default_params(rng_mode=oracle.RNG_PHILOX, **{k: v for k, v in ok.items() if k not in ("ransac_batch",)})
    if "ransac_batch" in ok:
        p_o.batch_size = ok["ransac_batch"]
    p_g = capi.default_params(**ok)
    return p_o, p_g


def test_philox_known_answer_vectors_on_the_device(lgr, oracle):
    """all three Random123 philox4x32-10 known-answer vectors through the device generator's FULL counter (the one function behind the RANSAC
    sampler and the closest-plane subsets), and the all-zero one through lgr_ransac_samples_dev: its triple is select3 of the vector's words"""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "philox_kat.json")))
    assert len(g["vectors"]) == 3
    for v in g["vectors"]:
        key = int(v["key"][0], 16) | int(v["key"][1], 16) << 32
        out = lgr.selfcheck_philox(key, [int(c, 16) for c in v["counter"]])
        assert [f"{x:08x}" for x in out] == v["out"]
    w = [int(x, 16) for x in g["vectors"][0]["out"]]
    for c in (5, 1000, 2 ** 30):
        got = lgr.ransac_samples(0, 0, 1, c).cpu().numpy()[0]
        np.testing.assert_array_equal(got, oracle.select3([x >> 1 for x in w[:3]], c))


def test_sampler_matches_oracle(lgr, oracle):
    for c in (3, 4, 17, 6000, 200000):
        got = lgr.ransac_samples(566, 1000, 5000, c).cpu().numpy()
        r = oracle.rng_stream(oracle.RNG_PHILOX, 566, 3 * 6000).reshape(-1, 3)[1000:6000]
        want = np.array([oracle.select3(x, c) for x in r[:700]])
        np.testing.assert_array_equal(got[:700], want)
        assert (got >= 0).all() and (got < c).all()
        # the reference's wrap-around branch (src/sac_prerejective_omp.cpp:60-65) can emit a duplicate index; it is
        # reproduced literally (such samples die in the polygon prerejection), so distinctness is only statistical
        if c >= 6000:
            distinct = (got[:, 0] != got[:, 1]) & (got[:, 1] != got[:, 2]) & (got[:, 0] != got[:, 2])
            assert distinct.mean() > 0.999


@pytest.mark.parametrize("metric,score", [(1, 2), (0, 0), (0, 1), (0, 2), (0, 3)])
def test_replay(lgr, oracle, problem, metric, score):
    from lgr_amd import capi
    p_o, p_g = params_pair(oracle, capi, metric_id=metric, score_id=score)
    corr = problem["corr"]
    triples = lgr.ransac_samples(566, 0, 4096, corr.shape[0])
    src, tgt = cuda(problem["src"]), cuda(problem["tgt"])
    ok, Ts, ninl, met = lgr.ransac_replay(src, tgt, corr, p_g, triples)
    ook, oTs, oninl, omet = oracle.replay(problem["src"], problem["tgt"], to_orc_corr(oracle, corr), p_o, triples.cpu().numpy())
    np.testing.assert_array_equal(ok, ook)
    assert ok.sum() > 20
    np.testing.assert_array_equal(bits(Ts), bits(oTs))
    np.testing.assert_array_equal(ninl, oninl)
    np.testing.assert_array_equal(bits(met), bits(omet))


def test_evaluate_mask_and_refit(lgr, oracle, problem):
    corr = problem["corr"]
    ocorr = to_orc_corr(oracle, corr)
    src, tgt = cuda(problem["src"]), cuda(problem["tgt"])
    T = problem["T_gt"].astype(np.float32)
    for metric, score in [(1, 2), (0, 2), (0, 3)]:
        mask, ni, rm, me = lgr.evaluate(src, tgt, corr, T, metric, score)
        omask, oni, orm, ome = oracle.evaluate(problem["src"], problem["tgt"], ocorr, T, metric, score)
        np.testing.assert_array_equal(mask, omask)
        assert ni == oni and abs(ni - 0.4 * len(corr)) < 0.05 * len(corr)
        assert np.float32(rm).view(np.uint32) == np.float32(orm).view(np.uint32)
        assert np.float32(me).view(np.uint32) == np.float32(ome).view(np.uint32)
    Tr = lgr.refit(src, tgt, corr, cuda(mask))
    oTr = oracle.refit(problem["src"], problem["tgt"], ocorr, omask)
    np.testing.assert_array_equal(bits(Tr), bits(oTr))
    assert np.abs(Tr - problem["T_gt"]).max() < 1e-3        # noise-limited
    R = Tr[:3, :3]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-5) and np.linalg.det(R) > 0.999


# small batches: the device evaluates up to 16 schedule batches per round of launches and replays them on the host; the
# result must be the batch-by-batch schedule's (the oracle's), also when the adaptive bound ends the loop inside a round
@pytest.mark.parametrize("metric,batch,iters", [(1, 4096, 20000), (0, 16384, 50000), (1, 1000, 3000), (1, 256, 20000), (0, 300, 30000), (0, 64, 12000)])
def test_ransac_whole_loop(lgr, oracle, problem, metric, batch, iters):
    from lgr_amd import capi
    p_o, p_g = params_pair(oracle, capi, metric_id=metric, score_id=2, max_iterations=iters, ransac_batch=batch)
    corr = problem["corr"]
    res, mask = lgr.ransac(cuda(problem["src"]), cuda(problem["tgt"]), corr, p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, corr), p_o)
    assert res.iterations == ores.iterations and res.best_iteration == ores.best_iteration
    assert res.num_rejections == ores.num_rejections and res.estimated_iters == ores.estimated_iters
    assert res.converged == ores.converged == 1
    assert res.n_inliers == ores.n_inliers
    np.testing.assert_array_equal(mask, omask)
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))
    assert np.float32(res.metric).view(np.uint32) == np.float32(ores.metric).view(np.uint32)
    assert np.abs(res.matrix() - problem["T_gt"]).max() < 1e-3
    assert res.iterations < iters or metric == 1    # the adaptive bound fires on this 40 % inlier problem


@pytest.mark.parametrize("metric,batch,iters", [(1, 1000, 20000), (0, 256, 30000), (1, 4096, 60000)])
def test_ransac_whole_loop_with_a_good_guess(lgr, oracle, problem, metric, batch, iters):
    """a guess (src/sac_prerejective_omp.cpp:134-147) sets the metric to beat but never enters the loop's record inlier set, so the
    candidate gate derived from that metric must not hide hypotheses that are records for the adaptive bound (:224-228): with a GOOD guess
    (the ground truth: nearly every hypothesis scores below it) iterations / estimated_iters must still be the oracle's, which applies no gate"""
    from lgr_amd import capi
    G = problem["T_gt"].astype(np.float32)
    p_o, p_g = params_pair(oracle, capi, metric_id=metric, score_id=2, max_iterations=iters, ransac_batch=batch, guess=G)
    corr = problem["corr"]
    res, mask = lgr.ransac(cuda(problem["src"]), cuda(problem["tgt"]), corr, p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, corr), p_o)
    assert (res.iterations, res.estimated_iters, res.num_rejections, res.best_iteration) == (ores.iterations, ores.estimated_iters, ores.num_rejections, ores.best_iteration)
    assert res.iterations < iters                          # the bound fired although (almost) nothing beats the guess's metric
    assert res.converged == ores.converged == 1 and res.n_inliers == ores.n_inliers
    np.testing.assert_array_equal(mask, omask)
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))


def test_ransac_degenerate(lgr, oracle, problem):
    from lgr_amd import capi
    p_o, p_g = params_pair(oracle, capi, metric_id=1, max_iterations=2000)
    corr = problem["corr"][:2]
    res, mask = lgr.ransac(cuda(problem["src"]), cuda(problem["tgt"]), corr, p_g)
    assert res.converged == 0 and np.array_equal(res.matrix(), np.eye(4, dtype=np.float32))
    # all-outlier correspondences: nothing reaches MIN_NR_INLIERS, result not converged on both sides
    rng = np.random.default_rng(0)
    bad = problem["corr"].copy()
    bad["index_match"] = rng.integers(0, problem["tgt"].shape[0], bad.shape[0])
    res, mask = lgr.ransac(cuda(problem["src"]), cuda(problem["tgt"]), bad, p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, bad), p_o)
    assert res.converged == ores.converged == 0
    assert res.iterations == ores.iterations


@pytest.fixture(scope="module")
def pair():
    from lgr_amd import synthetic
    return synthetic.make_pair(20000, seed=11)


@pytest.mark.parametrize("matching", [0, 1, 2])
def test_filters(lgr, oracle, pair, matching):
    rng = np.random.default_rng(matching)
    ns, nt = pair["src"].shape[0], pair["tgt"].shape[0]
    # spatially coherent 1-NN tables (geometric nearest neighbours under the ground truth), 25 % corrupted,
    # some unmatched (-1): exercises mutual and neighbourhood-consistency logic with a realistic keep rate
    T = pair["T_gt"]
    from lgr_amd.synthetic import make_points
    src_in_tgt = make_points(pair["src"][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3])
    tgt_in_src = make_points((pair["tgt"][:, :3].astype(np.float64) - T[:3, 3]) @ T[:3, :3])
    ij = oracle.knn(src_in_tgt, pair["tgt"], 1)[0][:, 0].astype(np.int32)
    ji = oracle.knn(tgt_in_src, pair["src"], 1)[0][:, 0].astype(np.int32)
    bad = rng.permutation(ns)[: ns // 4]; ij[bad] = rng.integers(0, nt, len(bad))
    bad = rng.permutation(nt)[: nt // 4]; ji[bad] = rng.integers(0, ns, len(bad))
    ij[rng.permutation(ns)[:50]] = -1
    ji[rng.permutation(nt)[:50]] = -1
    dij = rng.uniform(0, 50, ns).astype(np.float32); dji = rng.uniform(0, 50, nt).astype(np.float32)
    want = oracle.filter_matches(matching, pair["src"], pair["tgt"], ij, dij, ji, dji, 0.1, 40)
    got = lgr.filter(matching, cuda(pair["src"]), cuda(pair["tgt"]), cuda(ij), cuda(dij), cuda(ji), cuda(dji), 0.1, 40)
    assert len(got) == len(want) and len(got) > 100
    np.testing.assert_array_equal(got["index_query"], want["query"])
    np.testing.assert_array_equal(got["index_match"], want["match"])
    np.testing.assert_array_equal(bits(got["distance"]), bits(want["distance"]))
    np.testing.assert_array_equal(bits(got["threshold"]), bits(want["threshold"]))


@pytest.mark.parametrize("matching", [0, 2])
def test_align_end_to_end(lgr, oracle, pair, matching):
    """alignPointClouds on a 20k-point synthetic pair: correspondences and final inlier set identical to the oracle,
    final transform bit-equal to the oracle's and within the north-star tolerance band of the ground truth."""
    from lgr_amd import capi
    kw = dict(matching_id=matching, bf_block_size=200000, max_iterations=100000, distance_thr=0.1,
              vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    p_o = oracle.default_params(rng_mode=oracle.RNG_PHILOX, **kw)
    p_g = capi.default_params(**kw)
    ores, ocorr, _ = oracle.align(pair["src"], pair["tgt"], p_o)
    src, tgt = cuda(pair["src"]), cuda(pair["tgt"])
    corr = lgr.correspondences(src, tgt, p_g).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    assert len(corr) == len(ocorr) and len(corr) > 200
    np.testing.assert_array_equal(corr["index_query"], ocorr["query"])
    np.testing.assert_array_equal(corr["index_match"], ocorr["match"])
    np.testing.assert_array_equal(bits(corr["distance"]), bits(ocorr["distance"]))
    np.testing.assert_array_equal(bits(corr["threshold"]), bits(ocorr["threshold"]))
    res = lgr.align(src, tgt, p_g)
    assert res.n_correspondences == len(ocorr)
    assert res.converged == ores.converged == 1
    assert res.iterations == ores.iterations and res.n_inliers == ores.n_inliers
    assert np.abs(res.matrix() - ores.matrix()).max() <= 1e-4          # north-star tolerance ...
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))   # ... and in fact bit-equal
    assert np.abs(res.matrix() - pair["T_gt"]).max() < 2e-2            # noise-limited accuracy vs ground truth
    res_h = lgr.align_host(pair["src"], pair["tgt"], p_g)
    np.testing.assert_array_equal(bits(res_h.matrix()), bits(res.matrix()))


def test_choose_best_hypothesis(lgr, oracle, problem):
    """chooseBestHypothesis (src/hypotheses.cpp:50-129, decision part): the hypothesis with the most uniformly spread
    correspondence inliers wins; identity / -1 when no hypothesis has any inliers."""
    from lgr_amd import synthetic
    rng = np.random.default_rng(9)
    T = problem["T_gt"]
    tns = [synthetic.random_se3(rng), T, T @ np.array([[1, 0, 0, 0.02], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32), synthetic.random_se3(rng)]
    src, tgt = cuda(problem["src"]), cuda(problem["tgt"])
    oc = to_orc_corr(oracle, problem["corr"])
    bi, Tb, uni = lgr.choose_best_hypothesis(src, tgt, problem["corr"], tns)
    oi, oT, ouni = oracle.choose_best_hypothesis(problem["src"], problem["tgt"], oc, tns)
    assert bi == oi and bi in (1, 2)
    np.testing.assert_array_equal(bits(uni), bits(ouni))
    np.testing.assert_array_equal(bits(Tb), bits(oT))
    bi, Tb, _ = lgr.choose_best_hypothesis(src, tgt, problem["corr"], [tns[0], tns[3]])
    assert bi == -1 and np.array_equal(Tb, np.eye(4, dtype=np.float32))
    bi, Tb, _ = lgr.choose_best_hypothesis(src, tgt, problem["corr"], [])
    assert bi == -1


# ---- n_samples other than 3 (src/sac_prerejective_omp.cpp:33-77,105-108,220 are generic in it; every shipped config uses 3) ---------------
@pytest.mark.parametrize("n_samples", [4, 5, 8])
def test_sampler_matches_oracle_n_samples(lgr, oracle, n_samples):
    n, n_corr = 2000, 1000
    got = lgr.ransac_samples(566, 11, n, n_corr, n_samples=n_samples).cpu().numpy()
    for i in range(n):
        assert got[i].tolist() == oracle.select_n(oracle.philox_draws(566, 11 + i, n_samples), n_corr), i
    # the wrap-around branch: with as many correspondences as samples most draws collide
    got = lgr.ransac_samples(3, 0, 256, n_samples, n_samples=n_samples).cpu().numpy()
    for i in range(256):
        assert got[i].tolist() == oracle.select_n(oracle.philox_draws(3, i, n_samples), n_samples), i


def test_sampler_n_samples_3_is_the_triple_sampler(lgr):
    a = lgr.ransac_samples(566, 5, 4096, 777).cpu().numpy()
    b = lgr.ransac_samples(566, 5, 4096, 777, n_samples=3).cpu().numpy()
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("n_samples", [4, 6])
@pytest.mark.parametrize("metric", [1, 0])
def test_replay_n_samples(lgr, oracle, problem, n_samples, metric):
    from lgr_amd import capi
    p_o, p_g = params_pair(oracle, capi, metric_id=metric, score_id=2, n_samples=n_samples)
    corr = problem["corr"]
    c, n = corr.shape[0], 6000
    tup = lgr.ransac_samples(566, 0, n, c, n_samples=n_samples).cpu().numpy()
    # half of the rows drawn among the inliers only, so that whole tuples get through the polygon test
    ocorr = to_orc_corr(oracle, corr)
    omask, _, _, _ = oracle.evaluate(problem["src"], problem["tgt"], ocorr, problem["T_gt"].astype(np.float32), 0, 2)
    inl = np.flatnonzero(omask)
    rng = np.random.default_rng(1)
    tup[: n // 2] = np.sort(np.stack([rng.choice(inl, n_samples, replace=False) for _ in range(n // 2)]), axis=1).astype(np.int32)
    ok, Ts, ninl, met = lgr.ransac_replay(cuda(problem["src"]), cuda(problem["tgt"]), corr, p_g, cuda(tup))
    ook, oTs, oninl, omet = oracle.replay(problem["src"], problem["tgt"], ocorr, p_o, tup)
    np.testing.assert_array_equal(ok, ook)
    assert ok.sum() > n // 8
    np.testing.assert_array_equal(bits(Ts), bits(oTs))
    np.testing.assert_array_equal(ninl, oninl)
    np.testing.assert_array_equal(bits(met), bits(omet))
    good = (ninl > 0.3 * c)
    assert good.sum() > 100 and np.abs(Ts[good].reshape(-1, 4, 4).transpose(0, 2, 1) - problem["T_gt"]).max() < 0.05


@pytest.mark.parametrize("n_samples,metric,batch,iters", [(4, 1, 4096, 40000), (5, 0, 1000, 60000)])
def test_ransac_whole_loop_n_samples(lgr, oracle, problem, n_samples, metric, batch, iters):
    from lgr_amd import capi
    p_o, p_g = params_pair(oracle, capi, metric_id=metric, score_id=2, max_iterations=iters, ransac_batch=batch, n_samples=n_samples)
    corr = problem["corr"]
    res, mask = lgr.ransac(cuda(problem["src"]), cuda(problem["tgt"]), corr, p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, corr), p_o)
    assert res.iterations == ores.iterations and res.best_iteration == ores.best_iteration
    assert res.num_rejections == ores.num_rejections and res.estimated_iters == ores.estimated_iters
    assert res.converged == ores.converged == 1
    assert res.n_inliers == ores.n_inliers
    np.testing.assert_array_equal(mask, omask)
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))
    assert np.abs(res.matrix() - problem["T_gt"]).max() < 1e-3


def test_n_samples_outside_the_built_range_is_unsupported(lgr, problem):
    from lgr_amd import capi
    for ns in (2, 9):
        with pytest.raises(capi.LgrError, match="rc=-5"):
            lgr.ransac(cuda(problem["src"]), cuda(problem["tgt"]), problem["corr"], capi.default_params(n_samples=ns))


# ---- the resident kernel (lgr_ctx_options.ransac_schedule = 2): ONE launch for the whole loop; results are the launch chain's and the oracle's --------
@pytest.fixture()
def resident(lgr):
    from lgr_amd import capi
    lgr.set_options(ransac_schedule=capi.RANSAC_SCHEDULE_RESIDENT)
    yield lgr
    lgr.set_options()


def _same_result(a, b, ma, mb):
    for f in ("iterations", "best_iteration", "num_rejections", "estimated_iters", "converged", "n_inliers"):
        assert getattr(a, f) == getattr(b, f), f
    np.testing.assert_array_equal(bits(a.matrix()), bits(b.matrix()))
    assert np.float32(a.metric).view(np.uint32) == np.float32(b.metric).view(np.uint32)
    np.testing.assert_array_equal(ma, mb)


@pytest.mark.parametrize("metric,batch,iters,n_samples", [(1, 4096, 20000, 3), (0, 16384, 50000, 3), (1, 1000, 3000, 3), (1, 256, 20000, 3), (0, 300, 30000, 3),
                                                          (0, 64, 12000, 3), (1, 65536, 1000000, 3), (1, 4096, 40000, 4), (0, 1000, 60000, 5), (1, 7, 1, 3)])
def test_resident_kernel_whole_loop(lgr, oracle, problem, metric, batch, iters, n_samples):
    from lgr_amd import capi
    p_o, p_g = params_pair(oracle, capi, metric_id=metric, score_id=2, max_iterations=iters, ransac_batch=batch, n_samples=n_samples)
    corr = problem["corr"]
    src, tgt = cuda(problem["src"]), cuda(problem["tgt"])
    chain, cmask = lgr.ransac(src, tgt, corr, p_g)
    try:
        lgr.set_options(ransac_schedule=capi.RANSAC_SCHEDULE_RESIDENT)
        res, mask = lgr.ransac(src, tgt, corr, p_g)
        res2, mask2 = lgr.ransac(src, tgt, corr, p_g)     # (the barrier words start from the host's record every time)
    finally:
        lgr.set_options()
    _same_result(res, chain, mask, cmask)
    _same_result(res2, chain, mask2, cmask)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, corr), p_o)
    _same_result(res, ores, mask, omask)
    if iters > 1:
        assert res.converged == 1 and np.abs(res.matrix() - problem["T_gt"]).max() < 1e-3


def test_resident_kernel_guess_and_degenerate(resident, oracle, problem):
    from lgr_amd import capi
    lgr = resident
    corr = problem["corr"]
    src, tgt = cuda(problem["src"]), cuda(problem["tgt"])
    # a good guess (the gate of ADVICE r4): iterations / estimated_iters still the oracle's
    p_o, p_g = params_pair(oracle, capi, metric_id=1, score_id=2, max_iterations=20000, ransac_batch=1000, guess=problem["T_gt"].astype(np.float32))
    res, mask = lgr.ransac(src, tgt, corr, p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, corr), p_o)
    _same_result(res, ores, mask, omask)
    # nothing reaches MIN_NR_INLIERS: the loop runs to max_iterations and ends not converged
    p_o, p_g = params_pair(oracle, capi, metric_id=1, max_iterations=2000)
    rng = np.random.default_rng(0)
    bad = corr.copy()
    bad["index_match"] = rng.integers(0, problem["tgt"].shape[0], bad.shape[0])
    res, mask = lgr.ransac(src, tgt, bad, p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, bad), p_o)
    assert res.converged == ores.converged == 0 and res.iterations == ores.iterations and res.num_rejections == ores.num_rejections
    # fewer correspondences than samples: refused before any launch
    res, mask = lgr.ransac(src, tgt, corr[:2], p_g)
    assert res.converged == 0 and np.array_equal(res.matrix(), np.eye(4, dtype=np.float32))
    # exactly n_samples correspondences: C(3, 3) = 1 iteration
    p_o, p_g = params_pair(oracle, capi, metric_id=0, max_iterations=2000)
    res, mask = lgr.ransac(src, tgt, corr[:3], p_g)
    ores, omask = oracle.ransac(problem["src"], problem["tgt"], to_orc_corr(oracle, corr[:3]), p_o)
    assert res.iterations == ores.iterations == 1 and res.converged == ores.converged


def test_resident_kernel_plane_metrics_keep_the_chain(resident, problem):
    """the plane metrics' loop has launches of its own between the phases: the option is ignored for them (same result as without it)"""
    from lgr_amd import capi
    lgr = resident
    p = capi.default_params(metric_id=capi.METRIC_COMBINATION if hasattr(capi, "METRIC_COMBINATION") else 3, score_id=0, max_iterations=3000, ransac_batch=1000)
    src, tgt = cuda(problem["src"]), cuda(problem["tgt"])
    a, ma = lgr.ransac(src, tgt, problem["corr"], p)
    lgr.set_options()
    b, mb = lgr.ransac(src, tgt, problem["corr"], p)
    _same_result(a, b, ma, mb)
