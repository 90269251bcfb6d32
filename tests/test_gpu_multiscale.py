"""GPU parity: multi-scale matching (feature_radius unset -> include/matching.h:176-262 initialize,
:264-352 match_multiscale) vs the oracle: per-key-point radius levels from the 5-NN density, the down-sampling chain,
per-level FPFH and brute-force matches, the proximity vote.  Bar: correspondences and transforms bit-exact."""
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
    """Non-uniform density (two sampling rates), so that key points really fall on different radius levels."""
    from lgr_amd import synthetic
    a = synthetic.make_pair(24000, seed=31)
    rng = np.random.default_rng(5)
    out = dict(a)
    for name in ("src", "tgt"):
        pts = a[name]
        x = pts[:, 0]
        thin = x > np.median(x)
        keep = ~thin | (rng.random(len(pts)) < 0.3)      # the far half keeps 30 % of its points
        out[name] = np.ascontiguousarray(pts[keep])
    return out


def check_corr(corr, ocorr):
    assert len(corr) == len(ocorr)
    np.testing.assert_array_equal(corr["index_query"], ocorr["query"])
    np.testing.assert_array_equal(corr["index_match"], ocorr["match"])
    np.testing.assert_array_equal(bits(corr["distance"]), bits(ocorr["distance"]))
    np.testing.assert_array_equal(bits(corr["threshold"]), bits(ocorr["threshold"]))


@pytest.mark.parametrize("matching", [0, 1, 2])
def test_multiscale_correspondences(lgr, oracle, pair, matching):
    from lgr_amd import capi
    kw = dict(feature_radius=0.0, matching_id=matching, bf_block_size=200000, distance_thr=0.1, iss_radius_src=0.05, iss_radius_tgt=0.05,
              vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    ocorr, _ = oracle.correspondences(pair["src"], pair["tgt"], oracle.default_params(**kw))
    corr = lgr.correspondences(cuda(pair["src"]), cuda(pair["tgt"]), capi.default_params(**kw)).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    assert len(ocorr) > 100
    check_corr(corr, ocorr)


def test_multiscale_with_iss_and_align(lgr, oracle, pair):
    from lgr_amd import capi
    kw = dict(feature_radius=0.0, matching_id=0, bf_block_size=200000, distance_thr=0.1, keypoint_id=1, iss_radius_src=0.06,
              iss_radius_tgt=0.06, max_iterations=50000, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    ores, ocorr, _ = oracle.align(pair["src"], pair["tgt"], oracle.default_params(rng_mode=oracle.RNG_PHILOX, **kw))
    s, t = cuda(pair["src"]), cuda(pair["tgt"])
    corr = lgr.correspondences(s, t, capi.default_params(**kw)).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    assert len(ocorr) > 20
    check_corr(corr, ocorr)
    res = lgr.align(s, t, capi.default_params(**kw))
    assert res.n_correspondences == len(ocorr) and res.iterations == ores.iterations and res.n_inliers == ores.n_inliers
    np.testing.assert_array_equal(bits(res.matrix()), bits(ores.matrix()))


def test_multiscale_uses_several_levels(oracle, pair):
    """The fixture is built so that the radius levels differ between the dense and the thinned half."""
    idx, d2 = oracle.knn(pair["src"], pair["src"], 5)
    dens = np.sqrt(d2[:, 4])
    fr = np.sqrt((np.float32(352) * dens * dens).astype(np.float64) / np.pi).astype(np.float32)
    lv = np.floor(np.log2(fr)).astype(int)
    vals, cnt = np.unique(lv, return_counts=True)
    assert (cnt * 10 >= cnt.max()).sum() >= 2
