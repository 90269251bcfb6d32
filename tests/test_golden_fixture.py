"""tests/golden/patch2k.npz (made by tests/golden/make_fixtures.py): every stage's output on a 2k-point patch.
CPU: the current oracle reproduces the committed fixture bit for bit (a guard against drift of the canonical orders).
GPU: the HIP path reproduces the committed fixture through the C ABI, without consulting the oracle at all."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(HERE, "golden", "patch2k.npz"))


def same(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        return a.shape == b.shape and np.array_equal(a.astype(np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
    return a.shape == b.shape and np.array_equal(a, b)


def test_oracle_reproduces_fixture(oracle, fx):
    src, tgt = fx["src"], fx["tgt"]
    voxel, radius = float(fx["voxel"]), float(fx["radius"])
    assert same(oracle.downsample(src, voxel, oracle.ORDER_CANONICAL), fx["ds_canonical"])
    assert same(oracle.downsample(src, voxel, oracle.ORDER_LIBSTDCXX), fx["ds_libstdcxx"])
    surf = oracle.normals_knn(fx["ds_canonical"], 30, vp=fx["vp_src"])
    assert same(surf, fx["surf_normals"])
    assert same(oracle.spfh(surf, radius), fx["spfh"])
    assert same(oracle.fpfh(src, surf, radius), fx["fpfh"])
    for blk in (256, 100000):
        i, d = oracle.match_bf(fx["feat_src"], fx["feat_tgt"], blk)
        assert same(i, fx[f"match_idx_{blk}"]) and same(d, fx[f"match_dist_{blk}"])
    assert fx["match_idx_256"][5] == 1500 and fx["match_idx_100000"][5] == 40      # later block wins / lowest index wins
    assert same(oracle.smoothed_densities(src, 2), fx["dens_src"])
    assert same(oracle.iss_keypoints(src, 1.0), fx["iss_idx"])
    p = oracle.default_params(matching_id=oracle.MATCH_LR, feature_radius=radius, bf_block_size=256, distance_thr=1.0, vp_src=fx["vp_src"], vp_tgt=fx["vp_tgt"])
    corr, _ = oracle.correspondences(src, tgt, p)
    assert np.array_equal(corr, fx["corr"])
    rs, rt, rc = fx["r_src"], fx["r_tgt"], fx["r_corr"]
    ok, Ts, ninl, met = oracle.replay(rs, rt, rc, oracle.default_params(metric_id=oracle.METRIC_UNIFORMITY), fx["triples"])
    assert same(ok, fx["replay_ok"]) and same(Ts, fx["replay_T"]) and same(ninl, fx["replay_ninl"]) and same(met, fx["replay_metric"])
    mask, n_inl, rmse, metric = oracle.evaluate(rs, rt, rc, fx["r_T_gt"], oracle.METRIC_UNIFORMITY, oracle.SCORE_MSE)
    assert same(mask, fx["gt_mask"]) and n_inl == int(fx["gt_eval"][0]) and np.float32(metric) == np.float32(fx["gt_eval"][2])
    assert same(oracle.refit(rs, rt, rc, mask), fx["gt_refit"])
    res, _ = oracle.ransac(rs, rt, rc, oracle.default_params(rng_mode=oracle.RNG_PHILOX, metric_id=oracle.METRIC_UNIFORMITY, max_iterations=20000))
    assert same(res.matrix(), fx["ransac_T"])
    assert [res.iterations, res.converged, res.n_inliers, res.best_iteration, res.num_rejections] == fx["ransac_stats"].tolist()


@pytest.mark.gpu
def test_hip_reproduces_fixture(lgr, fx):
    import torch
    from lgr_amd import capi

    def cu(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()
    src, tgt = fx["src"], fx["tgt"]
    voxel, radius = float(fx["voxel"]), float(fx["radius"])
    assert same(lgr.downsample(cu(src), voxel).cpu().numpy(), fx["ds_canonical"])
    assert same(lgr.downsample_host(src, voxel, capi.ORDER_REFERENCE), fx["ds_libstdcxx"])
    surf = cu(fx["ds_canonical"].copy())
    lgr.normals_knn(surf, 30, None, fx["vp_src"])
    assert same(surf.cpu().numpy(), fx["surf_normals"])
    assert same(lgr.fpfh(cu(src), surf, radius).cpu().numpy(), fx["fpfh"])
    for blk in (256, 100000):
        i, d = lgr.match_bf(cu(fx["feat_src"]), cu(fx["feat_tgt"]), blk)
        assert same(i.cpu().numpy(), fx[f"match_idx_{blk}"])
        ok = fx[f"match_idx_{blk}"] >= 0
        assert same(d.cpu().numpy()[ok], fx[f"match_dist_{blk}"][ok])
    assert same(lgr.smoothed_densities(cu(src), 2).cpu().numpy(), fx["dens_src"])
    assert same(lgr.iss_keypoints(cu(src), 1.0).cpu().numpy(), fx["iss_idx"])
    p = capi.default_params(matching_id=0, feature_radius=radius, bf_block_size=256, distance_thr=1.0, vp_src=fx["vp_src"], vp_tgt=fx["vp_tgt"])
    corr = lgr.correspondences(cu(src), cu(tgt), p).cpu().numpy().view(capi.CORR_DTYPE).reshape(-1)
    oc = fx["corr"]
    assert np.array_equal(corr["index_query"], oc["query"]) and np.array_equal(corr["index_match"], oc["match"])
    assert same(corr["distance"], oc["distance"]) and same(corr["threshold"], oc["threshold"])
    rs, rt = cu(fx["r_src"]), cu(fx["r_tgt"])
    rc = np.zeros(len(fx["r_corr"]), capi.CORR_DTYPE)
    for a, b in (("index_query", "query"), ("index_match", "match"), ("distance", "distance"), ("threshold", "threshold")):
        rc[a] = fx["r_corr"][b]
    pr = capi.default_params(metric_id=capi.METRIC_UNIFORMITY)
    ok, Ts, ninl, met = lgr.ransac_replay(rs, rt, rc, pr, cu(fx["triples"]))
    assert same(ok, fx["replay_ok"])
    good = fx["replay_ok"] > 0
    assert same(Ts[good], fx["replay_T"][good]) and same(ninl[good], fx["replay_ninl"][good])
    cand = good & (fx["replay_ninl"] >= 10)          # the device evaluates the metric of candidates with >= 10 inliers only
    assert same(met[cand], fx["replay_metric"][cand])
    mask, n_inl, rmse, metric = lgr.evaluate(rs, rt, rc, fx["r_T_gt"], capi.METRIC_UNIFORMITY, capi.SCORE_MSE)
    assert same(mask, fx["gt_mask"]) and n_inl == int(fx["gt_eval"][0]) and np.float32(metric) == np.float32(fx["gt_eval"][2])
    assert same(lgr.refit(rs, rt, rc, cu(fx["gt_mask"])), fx["gt_refit"])
    res, _ = lgr.ransac(rs, rt, rc, capi.default_params(metric_id=capi.METRIC_UNIFORMITY, max_iterations=20000))
    assert same(res.matrix(), fx["ransac_T"])
    assert [res.iterations, res.converged, res.n_inliers, res.best_iteration, res.num_rejections] == fx["ransac_stats"].tolist()
