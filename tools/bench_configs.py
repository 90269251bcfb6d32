"""GPU measurements of the other BASELINE configs (not the bench line): config 4 (5M-point feature stage, RANSAC
stress with C = 200k at 60 % outliers, 1e5 fixed iterations) and config 5 (GROR, C = 50k at 60 % outliers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi, synthetic

ctx = capi.Context(0)
if os.environ.get("LGR_RANSAC_SCHEDULE"):   # 1 chain / 2 resident kernel (lgr_ctx_options.ransac_schedule)
    ctx.set_options(ransac_schedule=int(os.environ["LGR_RANSAC_SCHEDULE"]))


def timed(f, n=5):
    """median of n individually timed calls after a warm-up (a sporadic host stall of tens of ms -- allocator, GC -- lands in one call, not in the figure)"""
    f(); ctx.sync(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        r = f()
        ctx.sync(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2], r


which = sys.argv[1:] or ["ransac", "gror", "features5m"]
if "ransac" in which:
    pr = synthetic.make_correspondence_problem(n_pts=1_000_000, c=200_000, inlier_frac=0.4, sigma=0.01, thr=0.05, seed=566)
    src, tgt = torch.from_numpy(pr["src"]).cuda(), torch.from_numpy(pr["tgt"]).cuda()
    p = capi.default_params(max_iterations=100000, confidence=1.0, metric_id=capi.METRIC_UNIFORMITY, distance_thr=0.05)
    ms, (res, mask) = timed(lambda: ctx.ransac(src, tgt, pr["corr"], p))
    print(f"config 4 RANSAC stress: C=200000, 60% outliers, iterations run {res.iterations}, inliers {res.n_inliers}, "
          f"{ms:.1f} ms -> {res.iterations / ms * 1e3 / 1e6:.2f} M hypotheses/s, err {np.abs(res.matrix() - pr['T_gt']).max():.2e}", flush=True)
    p2 = capi.default_params(max_iterations=100000, metric_id=capi.METRIC_UNIFORMITY, distance_thr=0.05)
    ms, (res, mask) = timed(lambda: ctx.ransac(src, tgt, pr["corr"], p2))
    print(f"config 4 RANSAC latency (adaptive early-out): iterations {res.iterations}, {ms:.1f} ms", flush=True)
if "gror" in which:
    pr = synthetic.make_correspondence_problem(n_pts=2_000_000, c=50_000, inlier_frac=0.4, sigma=0.01, thr=0.05, seed=567)
    src, tgt = torch.from_numpy(pr["src"]).cuda(), torch.from_numpy(pr["tgt"]).cuda()
    ms, (res, mask) = timed(lambda: ctx.gror(src, tgt, pr["corr"], 0.05))
    print(f"config 5 GROR: C=50000, 60% outliers, K=800: {ms:.1f} ms, best_count {int(res.metric)}, refine inliers {res.n_inliers}, "
          f"err {np.abs(res.matrix() - pr['T_gt']).max():.2e}", flush=True)
if "features5m" in which:
    pair = synthetic.make_pair(5_000_000, seed=566)
    cloud = torch.from_numpy(pair["src"]).cuda()
    r = 0.25
    voxel = float(np.sqrt(np.pi * r * r / 352))
    ms_d, surf = timed(lambda: ctx.downsample(cloud, voxel))
    surf = surf.clone()
    ms_n, _ = timed(lambda: ctx.normals_knn(surf, 30, None, pair["vp_src"]))
    ms_f, f = timed(lambda: ctx.fpfh(cloud, surf, r))
    print(f"config 4 features at 5M points/cloud: surface {surf.shape[0]}, downsample {ms_d:.1f} ms, normals {ms_n:.1f} ms, FPFH {ms_f:.1f} ms", flush=True)
if "align5m" in which:
    pair = synthetic.make_pair(5_000_000, seed=566)
    s, t = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    p = capi.default_params(matching_id=0, metric_id=capi.METRIC_UNIFORMITY, feature_radius=0.25, bf_block_size=200000, max_iterations=1000000,
                            distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    ms, res = timed(lambda: ctx.align(s, t, p), n=1)
    print(f"5M-point pair end to end: {ms:.0f} ms, stage_ms {list(res.stage_ms)[:7]}, work fraction {ctx.match_work():.3f}, "
          f"err {np.abs(res.matrix() - pair['T_gt']).max():.2e}", flush=True)
if "iss1m" in which:
    pair = synthetic.make_pair(1_000_000, seed=566)
    s, t = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    for r in (0.05, 0.1):
        ms, idx = timed(lambda: ctx.iss_keypoints(s, r))
        print(f"ISS on 1M points, radius {r}: {idx.shape[0]} key points, {ms:.1f} ms", flush=True)
    p = capi.default_params(matching_id=0, metric_id=capi.METRIC_UNIFORMITY, feature_radius=0.25, bf_block_size=200000, max_iterations=1000000,
                            distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"], keypoint_id=1, iss_radius_src=0.05, iss_radius_tgt=0.05)
    ms, res = timed(lambda: ctx.align(s, t, p), n=5)
    print(f"1M-point pair with ISS key points (r = 0.05): {ms:.1f} ms, correspondences {res.n_correspondences}, inliers {res.n_inliers}, "
          f"converged {res.converged}, err {np.abs(res.matrix() - pair['T_gt']).max():.2e}", flush=True)
if "plane1m" in which:
    pair = synthetic.make_pair(1_000_000, seed=566)
    s, t = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    ctx.normals_knn(t, 30, None, pair["vp_tgt"])
    for mid, name in ((3, "combination"), (2, "closest_plane")):
        p = capi.default_params(matching_id=0, metric_id=mid, feature_radius=0.25, bf_block_size=200000, max_iterations=200000,
                                distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"], keypoint_id=1, iss_radius_src=0.05, iss_radius_tgt=0.05)
        ms, res = timed(lambda: ctx.align(s, t, p), n=5)
        print(f"1M-point pair, ISS key points, metric {name}: {ms:.1f} ms (RANSAC {list(res.stage_ms)[5]:.1f} ms, {res.iterations} iterations), "
              f"inliers {res.n_inliers}, metric {res.metric:.4f}, err {np.abs(res.matrix() - pair['T_gt']).max():.2e}", flush=True)
