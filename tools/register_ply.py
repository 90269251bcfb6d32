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
    ap.add_argument("--distance-thr", type=float, default=0.0, help="<= 0: automatic, 4 x the larger cloud density (src/common.cpp:267)")
    ap.add_argument("--iterations", type=int, default=1000000)
    ap.add_argument("--out", default=None, help="transformations.csv to append to")
    a = ap.parse_args()

    import numpy as np
    from lgr_amd import capi, formats, profile
    ctx = capi.Context(0)
    t = time.perf_counter()
    ld = profile.load_pair(ctx, a.source, a.target)        # loadPointClouds: duplicates, 2 x density voxel grid, normals
    ctx.sync()
    for side, path in (("src", a.source), ("tgt", a.target)):
        print(f"{os.path.basename(path)}: {len(ld['raw_' + side])} points -> {ld[side].shape[0]} after preprocessing "
              f"(voxel {ld['voxel_' + side]:.4g}, density {ld['density_' + side]:.4g})")
    print(f"loaded in {1e3 * (time.perf_counter() - t):.1f} ms")
    clouds = [ld["src"], ld["tgt"]]
    # getParametersFromConfig with the keys left out: distance_thr = 4 x max density, iss_radius = 2 x density (src/common.cpp:266-271,325-333)
    p = profile.default_profile(capi, ld["density_src"], ld["density_tgt"], keypoint=a.keypoint, metric=a.metric, matching=a.matching,
                                alignment=a.alignment, feature_radius=a.feature_radius if a.feature_radius > 0 else None,
                                distance_thr=a.distance_thr if a.distance_thr > 0 else None, iterations=a.iterations,
                                normals_available=ld["normals_available"])
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
