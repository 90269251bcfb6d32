"""Register two PLY scans end to end on one MI355X, the way the reference's `registration alignment` run does
(loadPointClouds -> alignPointClouds -> transformations.csv), through the C ABI:

    python tools/register_ply.py source.ply target.ply [--keypoint iss|any] [--metric uniformity|combination|...]
                                 [--feature-radius R] [--distance-thr D] [--out transformations.csv]

Steps: formats.read_ply (include/io.h) -> lgr_preprocess (duplicate filter, 2 x density voxel grid, normals;
src/common.cpp:429-470) -> lgr_align (src/alignment.cpp:72-109) -> formats.save_transformation (src/common.cpp:127-153).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source"); ap.add_argument("target")
    ap.add_argument("--keypoint", default="iss", choices=["iss", "any"])           # the reference's default (src/common.cpp:247)
    ap.add_argument("--metric", default="uniformity", choices=["uniformity", "correspondences", "closest_plane", "combination"])
    ap.add_argument("--matching", default="cluster", choices=["lr", "one_sided", "cluster"])
    ap.add_argument("--alignment", default="ransac", choices=["ransac", "gror"])
    ap.add_argument("--feature-radius", type=float, default=0.0, help="<= 0: multi-scale (the reference's behaviour when unset)")
    ap.add_argument("--distance-thr", type=float, default=0.0, help="<= 0: 2 x the coarser of the two voxel sizes")
    ap.add_argument("--iterations", type=int, default=1000000)
    ap.add_argument("--out", default=None, help="transformations.csv to append to")
    a = ap.parse_args()

    import numpy as np
    import torch
    from lgr_amd import capi, formats
    ctx = capi.Context(0)
    clouds, voxels = [], []
    for path in (a.source, a.target):
        pts, fields = formats.read_ply(path)
        t = time.perf_counter()
        out, voxel = ctx.preprocess(torch.from_numpy(pts).cuda(), normals_available=formats.has_normals(fields))
        ctx.sync()
        print(f"{os.path.basename(path)}: {len(pts)} points -> {out.shape[0]} after preprocessing (voxel {voxel:.4g}, {1e3 * (time.perf_counter() - t):.1f} ms)")
        clouds.append(out.clone()); voxels.append(voxel)
    thr = a.distance_thr if a.distance_thr > 0 else 2.0 * max(voxels)
    p = capi.default_params(
        keypoint_id=capi.KEYPOINT_ISS if a.keypoint == "iss" else capi.KEYPOINT_ANY,
        iss_radius_src=2.0 * voxels[0], iss_radius_tgt=2.0 * voxels[1],              # "automatic ISS radius": 2 x density (src/common.cpp:328)
        metric_id={"correspondences": 0, "uniformity": 1, "closest_plane": 2, "combination": 3}[a.metric],
        matching_id={"lr": 0, "one_sided": 1, "cluster": 2}[a.matching], alignment_id=1 if a.alignment == "gror" else 0,
        feature_radius=a.feature_radius, distance_thr=thr, bf_block_size=200000, max_iterations=a.iterations)
    t = time.perf_counter()
    res = ctx.align(clouds[0], clouds[1], p)
    dt = time.perf_counter() - t
    T = res.matrix()
    print(f"aligned in {1e3 * dt:.1f} ms: converged={res.converged} correspondences={res.n_correspondences} inliers={res.n_inliers} "
          f"metric={res.metric:.4f} iterations={res.iterations}")
    print(np.array2string(T, precision=6, suppress_small=True))
    if a.out:
        name = os.path.splitext(os.path.basename(a.source))[0] + "_" + os.path.splitext(os.path.basename(a.target))[0]
        formats.save_transformation(a.out, name, T)


if __name__ == "__main__":
    main()
