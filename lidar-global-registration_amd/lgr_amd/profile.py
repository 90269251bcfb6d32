"""The reference's YAML-default profile on top of the C ABI: what `getParametersFromConfig` (src/common.cpp:209-413) derives
from the two LOADED clouds when the config leaves a key out, and the loader steps in front of it (src/common.cpp:429-470).

Only the data-dependent defaults live here -- the YAML reader itself is out of scope:

  distance_thr   unset -> 4 * max(density_src, density_tgt)                    src/common.cpp:266-271
  iss_radius     unset -> 2 * density_src / 2 * density_tgt                    src/common.cpp:325-333
  density        = calculatePointCloudDensity(cloud) on the PREPROCESSED cloud src/common.cpp:226-227, :202-208
  keypoint iss, feature_radius unset (multi-scale), matching cluster, metric uniformity, score mse ...
                                                                               data/test.yaml:3-24 / src/common.cpp:247,273,335-413

`mod` is the binding the parameters are built for: lgr_amd.capi (the HIP path) or the test oracle's module -- both expose
default_params(**kw) with the same field names, which is what lets a parity test state the profile once.
"""
import numpy as np

MATCHING = {"lr": 0, "one_sided": 1, "cluster": 2}
METRIC = {"correspondences": 0, "uniformity": 1, "closest_plane": 2, "combination": 3}
SCORE = {"constant": 0, "mae": 1, "mse": 2, "exp": 3}
KEYPOINT = {"any": 0, "iss": 1}
ALIGNMENT = {"ransac": 0, "gror": 1}


def auto_thresholds(density_src, density_tgt):
    """(distance_thr, iss_radius_src, iss_radius_tgt) in the reference's float arithmetic (src/common.cpp:267,328-329)."""
    ds, dt = np.float32(density_src), np.float32(density_tgt)
    return float(np.float32(4) * max(ds, dt)), float(np.float32(2) * ds), float(np.float32(2) * dt)


def default_profile(mod, density_src, density_tgt, *, keypoint="iss", feature_radius=None, matching="cluster", metric="uniformity",
                    score="mse", alignment="ransac", distance_thr=None, iss_radius=None, iterations=1000000, block_size=200000,
                    normals_available=False, vp_src=None, vp_tgt=None, **extra):
    """Parameters of one `test:` entry of the reference's YAML (data/test.yaml:3-24) for clouds that went through the loader.
    `normals_available` is what alignment sees AFTER loadPointClouds: the loader has already estimated (or re-oriented) the normals,
    but `parameters.normals_available` still reports whether the FILES had them (src/common.cpp:228-229)."""
    thr, iss_s, iss_t = auto_thresholds(density_src, density_tgt)
    kw = dict(keypoint_id=KEYPOINT[keypoint], iss_radius_src=iss_s if iss_radius is None else float(iss_radius),
              iss_radius_tgt=iss_t if iss_radius is None else float(iss_radius),
              feature_radius=0.0 if feature_radius is None else float(feature_radius),          # <= 0: multi-scale (include/matching.h:176)
              feature_nr_points=352, normal_nr_points=30, scale_factor=2.0, cluster_k=40, bf_block_size=int(block_size),
              matching_id=MATCHING[matching], metric_id=METRIC[metric], score_id=SCORE[score],
              edge_thr_coef=0.95, confidence=0.999, max_iterations=int(iterations), n_samples=3,
              distance_thr=thr if distance_thr is None else float(distance_thr), normals_available=int(bool(normals_available)))
    if hasattr(mod, "ALIGN_GROR"):
        kw["alignment_id"] = ALIGNMENT[alignment]
    if vp_src is not None:
        kw["vp_src"] = vp_src
    if vp_tgt is not None:
        kw["vp_tgt"] = vp_tgt
    kw.update(extra)
    return mod.default_params(**kw)


def load_pair(ctx, src_path, tgt_path, vp_src=None, vp_tgt=None):
    """loadPointClouds (src/common.cpp:429-470) on the device: read both PLY files, filter duplicates, intensity = 1, voxel grid at
    2 x density, normals (k = 30).  -> dict(src, tgt: cuda tensors [n x 12]; voxel_*, density_*, normals_available, raw_*)"""
    import torch
    from . import formats
    out = {}
    avail = True
    raw = {}
    for side, path in (("src", src_path), ("tgt", tgt_path)):
        pts, fields = formats.read_ply(path)
        raw[side] = pts
        avail = avail and formats.has_normals(fields)
    out["normals_available"] = avail
    for side, vp in (("src", vp_src), ("tgt", vp_tgt)):
        cloud, voxel = ctx.preprocess(torch.from_numpy(raw[side]).cuda(ctx.device), vp=vp, normals_available=avail)
        out[side] = cloud.clone()
        out["voxel_" + side] = voxel
        out["density_" + side] = ctx.cloud_density(out[side])
        out["raw_" + side] = raw[side]
    return out
