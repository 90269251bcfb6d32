"""Generates tests/golden/patch2k.npz: a 2k-point synthetic patch with the ORACLE's outputs for every stage of the hot path
(SURVEY 8c item 3).  The oracle cannot be pinned against the reference here (it is unbuildable); this fixture
pins the oracle -- and through it the HIP path -- against accidental drift: tests/test_golden_fixture.py checks that the
current oracle reproduces it bit for bit (CPU) and that the HIP path does (GPU).

    python tests/golden/make_fixtures.py        # rewrites the fixture (only when a canonical order changes on purpose)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "lidar-global-registration_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as o  # noqa: E402
from lgr_amd import synthetic  # noqa: E402


def build():
    pair = synthetic.make_pair(2000, seed=97, constant_density=False)   # 2000 points over the full 24 m x 16 m scene
    src, tgt = pair["src"], pair["tgt"]
    out = dict(src=src, tgt=tgt, vp_src=np.asarray(pair["vp_src"], np.float32), vp_tgt=np.asarray(pair["vp_tgt"], np.float32))
    voxel = np.float32(0.35)
    out["voxel"] = voxel
    out["ds_canonical"] = o.downsample(src, float(voxel), o.ORDER_CANONICAL)
    out["ds_libstdcxx"] = o.downsample(src, float(voxel), o.ORDER_LIBSTDCXX)
    surf = o.normals_knn(out["ds_canonical"], 30, vp=pair["vp_src"])
    out["surf_normals"] = surf
    radius = 2.0
    out["radius"] = np.float32(radius)
    out["spfh"] = o.spfh(surf, radius)
    out["fpfh"] = o.fpfh(src, surf, radius)
    surf_t = o.normals_knn(o.downsample(tgt, float(voxel), o.ORDER_CANONICAL), 30, vp=pair["vp_tgt"])
    ft = o.fpfh(tgt, surf_t, radius)
    fs = out["fpfh"].copy()
    # engineered exact ties: inside one bf block (lowest index wins) and across blocks (later block wins)
    ft[40] = ft[700]; ft[1500] = ft[700]; fs[5] = ft[700]
    out["feat_src"], out["feat_tgt"] = fs, ft
    for blk in (256, 100000):
        i, d = o.match_bf(fs, ft, blk)
        out[f"match_idx_{blk}"], out[f"match_dist_{blk}"] = i, d
    out["dens_src"] = o.smoothed_densities(src, 2)
    out["iss_idx"] = o.iss_keypoints(src, 1.0)
    p = o.default_params(matching_id=o.MATCH_LR, feature_radius=radius, bf_block_size=256, distance_thr=1.0, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    corr, _ = o.correspondences(src, tgt, p)
    out["corr"] = corr
    # RANSAC pieces on directly synthesised correspondences (FPFH on a 2000-point cloud is too weak to give RANSAC work)
    pr_ = synthetic.make_correspondence_problem(n_pts=2000, c=600, inlier_frac=0.5, seed=98)
    rc = np.zeros(600, o.CORR_DTYPE)
    rc["query"] = pr_["corr"]["index_query"]; rc["match"] = pr_["corr"]["index_match"]
    rc["distance"] = pr_["corr"]["distance"]; rc["threshold"] = pr_["corr"]["threshold"]
    out["r_src"], out["r_tgt"], out["r_corr"], out["r_T_gt"] = pr_["src"], pr_["tgt"], rc, pr_["T_gt"].astype(np.float32)
    rng = np.random.default_rng(5)
    triples = np.stack([o.select3([int(x) for x in rng.integers(0, 2 ** 31 - 1, 3)], 600) for _ in range(256)]).astype(np.int32)
    out["triples"] = triples
    pr = o.default_params(metric_id=o.METRIC_UNIFORMITY)
    ok, Ts, ninl, met = o.replay(pr_["src"], pr_["tgt"], rc, pr, triples)
    out["replay_ok"], out["replay_T"], out["replay_ninl"], out["replay_metric"] = ok, Ts, ninl, met
    mask, n_inl, rmse, metric = o.evaluate(pr_["src"], pr_["tgt"], rc, pr_["T_gt"], o.METRIC_UNIFORMITY, o.SCORE_MSE)
    out["gt_mask"], out["gt_eval"] = mask, np.array([n_inl, rmse, metric], np.float64)
    out["gt_refit"] = o.refit(pr_["src"], pr_["tgt"], rc, mask)
    res, mask2 = o.ransac(pr_["src"], pr_["tgt"], rc, o.default_params(rng_mode=o.RNG_PHILOX, metric_id=o.METRIC_UNIFORMITY, max_iterations=20000))
    out["ransac_T"] = res.matrix()
    out["ransac_stats"] = np.array([res.iterations, res.converged, res.n_inliers, res.best_iteration, res.num_rejections], np.int64)
    out["T_gt"] = pair["T_gt"].astype(np.float32)
    return out


if __name__ == "__main__":
    fx = build()
    path = os.path.join(HERE, "patch2k.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path), "bytes;", len(fx["corr"]), "correspondences,", len(fx["iss_idx"]), "ISS key points")
