"""BASELINE-size parity (-m gpu): the 1M-point pair of BASELINE configs[1] (seed 566), every stage of the hot path checked
against the CPU ORACLE at that size -- not against the HIP path itself.

The oracle's voxel grid, k-NN normals, FPFH, density filter and RANSAC finish in seconds at 1M points on the box's cores, so
those stages are compared IN FULL (every row, bit for bit), each fed with the HIP path's output of the stage before (so a
difference is pinned to one stage).  Only the brute-force matcher (1e12 distance evaluations per direction) is sampled:
4096 random queries per direction against ALL 1M train rows through oracle.match_bf_subset (matchBF semantics,
include/matching.h:594-634: bf blocks of 200 000, later block wins a tie, lowest index inside a block, NaN rows skipped) --
index AND distance bits.  The staged chain is finally tied to the one-call pipeline (lgr_correspondences_dev / lgr_align_dev):
same correspondences, same transform."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0.25
# include/matching.h:231 `sqrtf(M_PI * search_radius * search_radius / (float) feature_nr_points)`: the product in double, rounded
# to float, square root in float (python floats on purpose: NumPy 2 would evaluate np.pi * np.float32 in float32)
VOXEL = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
assert VOXEL.hex() == "0x1.82f52e0000000p-6"
BLOCK = 200000
N_SAMPLE = 4096


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def st(lgr):
    """HIP-side stages of the 1M pair, computed once; tensors stay on the device, host copies beside them."""
    import torch
    from lgr_amd import synthetic
    pair = synthetic.make_pair(1_000_000, seed=566)
    s = dict(pair=pair)
    for side in ("src", "tgt"):
        cloud = torch.from_numpy(pair[side]).cuda()
        surf = lgr.downsample(cloud, VOXEL).clone()
        nrm = lgr.normals_knn(surf.clone(), 30, vp=pair["vp_" + side])       # in place on its argument
        feat = lgr.fpfh(cloud, nrm, float(R))
        lgr.sync()
        s[side] = dict(cloud=cloud, surf=surf, nrm=nrm, feat=feat, surf_h=surf.cpu().numpy(), nrm_h=nrm.cpu().numpy(), feat_h=feat.cpu().numpy())
    return s


@pytest.mark.parametrize("side", ["src", "tgt"])
def test_downsample_1m_full(st, oracle, side):
    want = oracle.downsample(st["pair"][side], VOXEL)
    got = st[side]["surf_h"]
    assert got.shape == want.shape and got.shape[0] > 500_000
    np.testing.assert_array_equal(bits(got), bits(want))


@pytest.mark.parametrize("side", ["src", "tgt"])
def test_normals_1m_full(st, oracle, side):
    want = oracle.normals_knn(st[side]["surf_h"], 30, vp=st["pair"]["vp_" + side])
    np.testing.assert_array_equal(bits(st[side]["nrm_h"]), bits(want))
    n = want[:, 4:7]
    assert np.isfinite(n).all() and np.abs(np.linalg.norm(n, axis=1) - 1).max() < 1e-5


@pytest.mark.parametrize("side", ["src", "tgt"])
def test_fpfh_1m_full(st, oracle, side):
    want = oracle.fpfh(st["pair"][side], st[side]["nrm_h"], float(R))
    got = st[side]["feat_h"]
    assert got.shape == (1_000_000, 33)
    np.testing.assert_array_equal(bits(got), bits(want))       # NaN rows included: same bit patterns
    ok = ~np.isnan(want).any(1)
    assert ok.mean() > 0.99 and np.abs(want[ok].reshape(-1, 3, 11).sum(2) - 100).max() < 1e-2


@pytest.fixture(scope="module")
def matches(lgr, st):
    ab_i, ab_d, ba_i, ba_d = lgr.match_bf2(st["src"]["feat"], st["tgt"]["feat"], BLOCK)
    lgr.sync()
    stats = lgr.match_stats()
    return dict(t=(ab_i, ab_d, ba_i, ba_d), h=[x.cpu().numpy() for x in (ab_i, ab_d, ba_i, ba_d)], stats=stats,
                work=lgr.match_work(), fmt=lgr.match_format())


def test_match_1m_sampled_oracle(st, matches, oracle):
    """the pruned / coarse-rejecting / re-filtered matcher on the pipeline's REAL FPFH rows vs the oracle's exhaustive scan."""
    assert matches["fmt"] == "f16r" and matches["work"] < 0.5            # the production schedule really ran (not a dense fallback)
    assert matches["stats"]["dense_ab"] == 0 and matches["stats"]["dense_ba"] == 0
    fs, ft = st["src"]["feat_h"], st["tgt"]["feat_h"]
    ab_i, ab_d, ba_i, ba_d = matches["h"]
    rng = np.random.default_rng(20261004)
    mism = 0
    for q, t, gi, gd in ((fs, ft, ab_i, ab_d), (ft, fs, ba_i, ba_d)):
        sel = np.sort(rng.choice(q.shape[0], N_SAMPLE, replace=False)).astype(np.int32)
        # always include some NaN query rows if there are any (no match: index -1)
        nan_rows = np.flatnonzero(np.isnan(q).any(1))[:64].astype(np.int32)
        sel = np.unique(np.concatenate([sel, nan_rows])).astype(np.int32)
        oi, od = oracle.match_bf_subset(q, sel, t, BLOCK)
        valid = oi >= 0
        mism += int((gi[sel] != oi).sum()) + int((bits(gd[sel])[valid] != bits(od)[valid]).sum())
        np.testing.assert_array_equal(gi[sel], oi)
        np.testing.assert_array_equal(bits(gd[sel])[valid], bits(od)[valid])
        assert (gi[nan_rows] == -1).all()
    assert mism == 0


def test_match_1m_full_oracle_one_direction(st, matches, oracle):
    """EVERY src -> tgt query of the 1M pair (1 000 000 queries x 1 000 000 train rows = 1e12 exact distances, about
    three minutes of oracle time on the box's host cores) against oracle.match_bf_subset -- index and distance bits of the production
    schedule's table.  (The other direction + this one = tools/full_match_check.py, profiles/r3_full_match_check_1M.json.)"""
    fs, ft = st["src"]["feat_h"], st["tgt"]["feat_h"]
    gi, gd = matches["h"][0], matches["h"][1]
    n = fs.shape[0]
    bad_i = bad_d = 0
    for lo in range(0, n, 100_000):
        sel = np.arange(lo, min(n, lo + 100_000), dtype=np.int32)
        oi, od = oracle.match_bf_subset(fs, sel, ft, BLOCK)
        ok = oi >= 0
        bad_i += int((gi[sel] != oi).sum())
        bad_d += int((bits(gd[sel])[ok] != bits(od)[ok]).sum())
    assert (bad_i, bad_d) == (0, 0)


def test_match_1m_refilter_on_off_identical(lgr, st, matches):
    """the rerank's MFMA re-filter + pair path (default) vs the whole-group exact scan, at BASELINE size."""
    lgr.set_match_options(rerank_refilter=0)
    try:
        off = [x.cpu().numpy() for x in lgr.match_bf2(st["src"]["feat"], st["tgt"]["feat"], BLOCK)]
        lgr.sync()
    finally:
        lgr.set_match_options()
    assert lgr.match_pairs() == (0, 0) or sum(lgr.match_pairs()) == 0
    for a, b in zip(matches["h"], off):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_match_1m_prune_off_identical(lgr, st, matches):
    """the dense schedule (prune = 0: every 32 x 32 tile computed, no bounds, no masks, no coarse rejection) vs the
    production schedule on the pair's REAL FPFH rows: all four tables bit-equal for ALL 2 M queries."""
    lgr.set_match_options(prune=0)
    try:
        dense = [x.cpu().numpy() for x in lgr.match_bf2(st["src"]["feat"], st["tgt"]["feat"], BLOCK)]
        lgr.sync()
        assert lgr.match_work() == 1.0
    finally:
        lgr.set_match_options()
    for a, b in zip(matches["h"], dense):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def corr(lgr, st, matches):
    from lgr_amd import capi
    ab_i, ab_d, ba_i, ba_d = matches["t"]
    return lgr.filter(capi.MATCH_LR, st["src"]["cloud"], st["tgt"]["cloud"], ab_i, ab_d, ba_i, ba_d, 0.1)


def test_filter_1m_full(st, matches, corr, oracle):
    ab_i, ab_d, ba_i, ba_d = matches["h"]
    want = oracle.filter_matches(oracle.MATCH_LR, st["pair"]["src"], st["pair"]["tgt"], ab_i, ab_d, ba_i, ba_d, 0.1)
    assert len(corr) == len(want) > 10000
    np.testing.assert_array_equal(corr["index_query"], want["query"])
    np.testing.assert_array_equal(corr["index_match"], want["match"])
    np.testing.assert_array_equal(bits(corr["distance"]), bits(want["distance"]))
    np.testing.assert_array_equal(bits(corr["threshold"]), bits(want["threshold"]))


def _params(mod, pair, **kw):
    return mod.default_params(matching_id=mod.MATCH_LR, metric_id=mod.METRIC_UNIFORMITY, score_id=mod.SCORE_MSE, feature_radius=0.25,
                              feature_nr_points=352, normal_nr_points=30, bf_block_size=BLOCK, edge_thr_coef=0.95, confidence=0.999,
                              max_iterations=1000000, distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"], **kw)


def test_ransac_1m_full_and_pipeline(lgr, st, corr, oracle):
    """RANSAC + refit on the 1M pair's correspondences vs the oracle (whole loop, Philox schedule on both sides), and the
    one-call pipeline (what bench.py times) == the staged chain."""
    from lgr_amd import capi
    pair = st["pair"]
    p_g = _params(capi, pair)
    res, mask = lgr.ransac(st["src"]["cloud"], st["tgt"]["cloud"], corr, p_g)
    oc = np.zeros(len(corr), oracle.CORR_DTYPE)
    for a, b in (("query", "index_query"), ("match", "index_match"), ("distance", "distance"), ("threshold", "threshold")):
        oc[a] = corr[b]
    ores, omask = oracle.ransac(pair["src"], pair["tgt"], oc, _params(oracle, pair, rng_mode=oracle.RNG_PHILOX))
    assert (res.iterations, res.n_inliers, res.best_iteration, res.converged) == (ores.iterations, ores.n_inliers, ores.best_iteration, ores.converged)
    np.testing.assert_array_equal(mask, omask)
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))
    assert np.abs(res.matrix().astype(np.float64) - ores.matrix().astype(np.float64)).max() <= 1e-4     # north-star tolerance on the 4x4
    # one-call pipeline == staged chain
    c2 = lgr.correspondences(st["src"]["cloud"], st["tgt"]["cloud"], p_g).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    assert len(c2) == len(corr)
    for f in ("index_query", "index_match"):
        np.testing.assert_array_equal(c2[f], corr[f])
    np.testing.assert_array_equal(bits(c2["distance"]), bits(corr["distance"]))
    full = lgr.align(st["src"]["cloud"], st["tgt"]["cloud"], p_g)
    np.testing.assert_array_equal(bits(full.matrix()), bits(res.matrix()))
    assert (full.iterations, full.n_inliers, full.n_correspondences) == (res.iterations, res.n_inliers, len(corr))


def test_cluster_1m_full(lgr, st, matches, oracle):
    """SURVEY 8(d)'s second run of config 2 -- `matching: cluster`, the YAML default (include/matching.h:480-551) -- at
    BASELINE size: the cluster filter in full (two 40-NN tables of 1M points, both directional consistency distances) and the whole
    RANSAC + refit on its ~3e5 correspondences vs the oracle; then the one-call pipeline == the staged chain."""
    from lgr_amd import capi
    pair = st["pair"]
    ab_i, ab_d, ba_i, ba_d = matches["t"]
    got = lgr.filter(capi.MATCH_CLUSTER, st["src"]["cloud"], st["tgt"]["cloud"], ab_i, ab_d, ba_i, ba_d, 0.1, 40)
    h = matches["h"]
    want = oracle.filter_matches(oracle.MATCH_CLUSTER, pair["src"], pair["tgt"], h[0], h[1], h[2], h[3], 0.1, 40)
    assert len(got) == len(want) > 100_000
    np.testing.assert_array_equal(got["index_query"], want["query"])
    np.testing.assert_array_equal(got["index_match"], want["match"])
    np.testing.assert_array_equal(bits(got["distance"]), bits(want["distance"]))
    np.testing.assert_array_equal(bits(got["threshold"]), bits(want["threshold"]))
    p_g = _params(capi, pair)
    p_g.matching_id = capi.MATCH_CLUSTER
    res, mask = lgr.ransac(st["src"]["cloud"], st["tgt"]["cloud"], got, p_g)
    p_o = _params(oracle, pair, rng_mode=oracle.RNG_PHILOX)
    p_o.matching_id = oracle.MATCH_CLUSTER
    ores, omask = oracle.ransac(pair["src"], pair["tgt"], want, p_o)
    assert (res.iterations, res.n_inliers, res.best_iteration, res.converged, res.num_rejections) == \
           (ores.iterations, ores.n_inliers, ores.best_iteration, ores.converged, ores.num_rejections)
    np.testing.assert_array_equal(mask, omask)
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))
    assert np.float32(res.metric) == np.float32(ores.metric)
    full = lgr.align(st["src"]["cloud"], st["tgt"]["cloud"], p_g)
    np.testing.assert_array_equal(bits(full.matrix()), bits(res.matrix()))
    assert (full.iterations, full.n_inliers, full.n_correspondences) == (res.iterations, res.n_inliers, len(got))


# ---------------------------------------------------------------------------------------------------------------------------------
# PCL 1.12.1's own arithmetic at BASELINE size: lgr_ctx_options.arithmetic = LGR_ARITH_PCL vs the oracle's ORC_ARITH_PCL.
# Normals and pair features are PCL's sequences in BOTH modes (test_normals_1m_full / test_fpfh_1m_full above compare them with the
# oracle's default, which is eigen33 + glibc 2.35's acosf / atan2f); the mode adds weightPointSPFHSignature's order and rounding steps.
@pytest.fixture(scope="module")
def pcl_feats(st):
    import torch
    from lgr_amd import capi
    ctx = capi.Context(0)
    ctx.set_options(arithmetic=capi.ARITH_PCL)
    out = {}
    for side in ("src", "tgt"):
        f = ctx.fpfh(st[side]["cloud"], st[side]["nrm"], float(R))
        ctx.sync()
        out[side] = dict(feat=f, feat_h=f.cpu().numpy())
    ctx.close()
    return out


@pytest.mark.parametrize("side", ["src", "tgt"])
def test_fpfh_1m_full_pcl_arithmetic(st, pcl_feats, oracle, side):
    oracle.set_arith_mode(oracle.ARITH_PCL)
    try:
        want = oracle.fpfh(st["pair"][side], st[side]["nrm_h"], float(R))
    finally:
        oracle.set_arith_mode(oracle.ARITH_CANONICAL)
    got = pcl_feats[side]["feat_h"]
    np.testing.assert_array_equal(bits(got), bits(want))       # 33 M values per cloud, NaN rows included
    fast = st[side]["feat_h"]
    ok = ~np.isnan(want).any(1)
    # the default mode's weighting differs from PCL's at rounding level only
    assert (bits(fast[ok]) != bits(want[ok])).any(1).mean() > 0.9 and np.abs(fast[ok] - want[ok]).max() < 2e-3


def test_match_1m_pcl_arithmetic_rows_through_the_existing_matcher(lgr, pcl_feats, oracle):
    """the match tables of the PCL-arithmetic FPFH rows: the exact matcher is indifferent to where its rows come from; sampled queries of
    both directions against the oracle's exhaustive scan (index and distance bits)"""
    ab_i, ab_d, ba_i, ba_d = [x.cpu().numpy() for x in lgr.match_bf2(pcl_feats["src"]["feat"], pcl_feats["tgt"]["feat"], BLOCK)]
    lgr.sync()
    fs, ft = pcl_feats["src"]["feat_h"], pcl_feats["tgt"]["feat_h"]
    rng = np.random.default_rng(20261005)
    for q, t, gi, gd in ((fs, ft, ab_i, ab_d), (ft, fs, ba_i, ba_d)):
        sel = np.sort(rng.choice(q.shape[0], N_SAMPLE, replace=False)).astype(np.int32)
        oi, od = oracle.match_bf_subset(q, sel, t, BLOCK)
        valid = oi >= 0
        np.testing.assert_array_equal(gi[sel], oi)
        np.testing.assert_array_equal(bits(gd[sel])[valid], bits(od)[valid])


def test_match_1m_planar_scene_irregular_rows_vs_oracle(lgr, oracle):
    """Round 5: the planar-dominated 1M pair (85 % of the points on 41 rectangles; the shape of the reference's TLS configs, data/tests.yaml).
    Its FPFH rows hold all-zero rows of isolated points, which take the matcher's exact side scan instead of costing the pair the rotated
    operand format (lgr_match_options.irregular_rows).  EVERY irregular row of both sides and 2048 sampled regular queries per direction
    against the oracle's exhaustive scan over all 1M train rows -- index and distance bits; the lane switched off gives the same tables."""
    import torch
    from lgr_amd import synthetic
    pair = synthetic.make_planar_pair(1_000_000, seed=566)
    feats = []
    for side in ("src", "tgt"):
        cloud = torch.from_numpy(pair[side]).cuda()
        surf = lgr.downsample(cloud, VOXEL).clone()
        feats.append(lgr.fpfh(cloud, lgr.normals_knn(surf.clone(), 30, vp=pair["vp_" + side]), float(R)))
    lgr.sync()
    on = [x.cpu().numpy() for x in lgr.match_bf2(feats[0], feats[1], BLOCK)]
    lgr.sync()
    n_a, n_b, gave_up = lgr.match_irregular()
    assert gave_up == 0 and n_a > 0 and n_b > 0, (n_a, n_b, gave_up)
    assert lgr.match_format() == "f16r"
    h = [f.cpu().numpy() for f in feats]
    rng = np.random.default_rng(5)
    for q, t, gi, gd, n_irr in ((h[0], h[1], on[0], on[1], n_a), (h[1], h[0], on[2], on[3], n_b)):
        fin = np.isfinite(q).all(1)
        irregular = np.flatnonzero(fin & (np.abs(q.reshape(-1, 3, 11).astype(np.float64).sum(2) - 100.0) > 1e-3).any(1))
        assert len(irregular) == n_irr
        sel = np.unique(np.concatenate([irregular, rng.choice(q.shape[0], 2048, replace=False)])).astype(np.int32)
        oi, od = oracle.match_bf_subset(q, sel, t, BLOCK)
        ok = oi >= 0
        assert int((gi[sel] != oi).sum()) == 0
        assert int((bits(gd[sel])[ok] != bits(od)[ok]).sum()) == 0
    lgr.set_match_options(irregular_rows=0)
    try:
        off = [x.cpu().numpy() for x in lgr.match_bf2(feats[0], feats[1], BLOCK)]
        lgr.sync()
        assert lgr.match_format() == "f16" and lgr.match_irregular() == (0, 0, 0)
    finally:
        lgr.set_match_options()
    for a, b in zip(on, off):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
