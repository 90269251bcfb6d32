"""What the oracle's documented deviations from PCL 1.12.1's arithmetic do to the north-star observables.

The CPU oracle runs the BASELINE configs[1] profile twice on the same synthetic pair -- ARITH_CANONICAL (the orders the HIP path
restates bit for bit) and ARITH_PCL (PCL's own: eigen33 closed-form normals, glibc 2.35's acosf swap test and atan2f, FPFH neighbours by
ascending distance with val = hist * w rounded, float bin adds, double block sums; --by-piece: each of the six pieces alone) -- and reports, stage by stage:
normals that differ in any bit / by more than 1e-5, FPFH rows that differ in any bit and the largest element difference, match
indices that differ (both directions), correspondences that differ, and max |dT| of the final 4x4 after RANSAC + refit.

    python tools/pcl_order_report.py --points 100000                 # CPU only (brute-force matching by the oracle)
    python tools/pcl_order_report.py --points 1000000 --gpu-matcher  # on the GPU box: the exact matcher of liblgr_hip.so does the two
                                                                     # 1M x 1M matchings (bit-identical to the oracle's, tools/full_match_check.py)
Prints one JSON line; tests/test_oracle_pcl_order.py asserts the 100 k figures, DESIGN.md section 6 tables both sizes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def run_mode(o, pair, mode, matcher, matching_id):
    """downsample -> normals -> FPFH -> match both ways -> filter -> RANSAC, all by the oracle (matcher: oracle or the GPU's exact one)"""
    o.set_arith_mode(mode)
    r = 0.25
    voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))
    out = {}
    feats = []
    for side in ("src", "tgt"):
        surf = o.downsample(pair[side], voxel)
        nrm = o.normals_knn(surf, 30, vp=pair["vp_" + side])
        f = o.fpfh(pair[side], nrm, r)
        out["nrm_" + side] = nrm
        out["feat_" + side] = f
        feats.append(f)
    ab_i, ab_d, ba_i, ba_d = matcher(feats[0], feats[1])
    out["match"] = (ab_i, ab_d, ba_i, ba_d)
    corr = o.filter_matches(matching_id, pair["src"], pair["tgt"], ab_i, ab_d, ba_i, ba_d, 0.1)
    out["corr"] = corr
    p = o.default_params(rng_mode=o.RNG_PHILOX, metric_id=o.METRIC_UNIFORMITY, score_id=o.SCORE_MSE, max_iterations=1000000,
                         distance_thr=0.1, edge_thr_coef=0.95, confidence=0.999, matching_id=matching_id)
    res, mask = o.ransac(pair["src"], pair["tgt"], corr, p)
    out["res"], out["mask"] = res, mask
    o.set_arith_mode(o.ARITH_CANONICAL)
    return out


def compare(a, b, pair):
    rep = {}
    for side in ("src", "tgt"):
        na, nb = a["nrm_" + side], b["nrm_" + side]
        d = np.abs(na[:, 4:7].astype(np.float64) - nb[:, 4:7].astype(np.float64))
        rep["normals_" + side] = {"points": int(len(na)), "differ_in_any_bit": int((bits(na[:, 4:7]) != bits(nb[:, 4:7])).any(1).sum()),
                                  "max_abs_component_diff": float(np.nanmax(d)), "differ_by_more_than_1e-5": int((d > 1e-5).any(1).sum()),
                                  "max_abs_curvature_diff": float(np.nanmax(np.abs(na[:, 9].astype(np.float64) - nb[:, 9])))}
        fa, fb = a["feat_" + side], b["feat_" + side]
        both = ~(np.isnan(fa).any(1) | np.isnan(fb).any(1))
        dd = np.abs(fa[both].astype(np.float64) - fb[both].astype(np.float64))
        rep["fpfh_" + side] = {"rows": int(len(fa)), "rows_differ_in_any_bit": int((bits(fa) != bits(fb)).any(1).sum()),
                               "max_abs_bin_diff": float(dd.max()), "mean_abs_bin_diff": float(dd.mean()),
                               "rows_with_a_bin_moved_by_more_than_0.5": int((dd > 0.5).any(1).sum()),
                               "nan_rows_differ": int((np.isnan(fa).any(1) != np.isnan(fb).any(1)).sum())}
    ma, mb = a["match"], b["match"]
    rep["match"] = {"src_to_tgt_indices_differ": int((ma[0] != mb[0]).sum()), "tgt_to_src_indices_differ": int((ma[2] != mb[2]).sum()),
                    "queries": int(len(ma[0]) + len(ma[2]))}
    ca, cb = a["corr"], b["corr"]
    sa = set(zip(ca["query"].tolist(), ca["match"].tolist())); sb = set(zip(cb["query"].tolist(), cb["match"].tolist()))
    rep["correspondences"] = {"canonical": len(sa), "pcl_order": len(sb), "in_both": len(sa & sb), "only_canonical": len(sa - sb), "only_pcl_order": len(sb - sa)}
    Ta, Tb = a["res"].matrix().astype(np.float64), b["res"].matrix().astype(np.float64)
    rep["ransac"] = {"canonical": {"iterations": int(a["res"].iterations), "inliers": int(a["res"].n_inliers), "converged": int(a["res"].converged)},
                     "pcl_order": {"iterations": int(b["res"].iterations), "inliers": int(b["res"].n_inliers), "converged": int(b["res"].converged)},
                     "max_abs_dT": float(np.abs(Ta - Tb).max()), "max_abs_dT_rotation": float(np.abs(Ta[:3, :3] - Tb[:3, :3]).max()),
                     "max_abs_dT_translation": float(np.abs(Ta[:3, 3] - Tb[:3, 3]).max()),
                     "max_abs_err_vs_ground_truth": {"canonical": float(np.abs(Ta - pair["T_gt"]).max()), "pcl_order": float(np.abs(Tb - pair["T_gt"]).max())}}
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--seed", type=int, default=566)
    ap.add_argument("--matching", default="lr", choices=["lr", "cluster"])
    ap.add_argument("--gpu-matcher", action="store_true")
    ap.add_argument("--mode", type=int, default=63, help="bit mask of PCL-arithmetic pieces (oracle.ARITH_PIECES): 1 eigen33 normals, 2 acosf swap test, 4 weighting neighbour "
                    "order, 8 weighting rounded product, 16 weighting running normaliser, 32 atan2f; 63 = all")
    ap.add_argument("--base", type=int, default=None, help="arithmetic mask of the run everything is compared WITH: default oracle.ARITH_CANONICAL (35: what the HIP library's default "
                    "mode restates since round 5); 0 = the canonical orders of rounds 1-4.  --by-piece always uses 0")
    ap.add_argument("--by-piece", action="store_true", help="every piece switched ALONE against the canonical run, then all together: one JSON with a summary row per piece")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import oracle as o
    o.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    o.set_num_threads(cores)
    from lgr_amd import synthetic
    pair = synthetic.make_pair(a.points, seed=a.seed)
    if a.gpu_matcher:
        import torch
        from lgr_amd import capi
        ctx = capi.Context(0)

        def matcher(fa, fb):
            out = [x.cpu().numpy() for x in ctx.match_bf2(torch.from_numpy(fa).cuda(), torch.from_numpy(fb).cuda(), 200000)]
            ctx.sync()
            return out
    else:
        def matcher(fa, fb):
            ab = o.match_bf(fa, fb, 200000); ba = o.match_bf(fb, fa, 200000)
            return ab[0], ab[1], ba[0], ba[1]
    mid = o.MATCH_LR if a.matching == "lr" else o.MATCH_CLUSTER
    t0 = time.time()
    base = 0 if a.by_piece else (o.ARITH_CANONICAL if a.base is None else a.base)
    can = run_mode(o, pair, base, matcher, mid)
    rep = {"workload": "BASELINE configs[1] profile, %d points per cloud, seed %d, matching %s" % (a.points, a.seed, a.matching),
           "libm": "glibc 2.35 float routines restated (oracle/src/orc_libm.h)", "base_mode_bits": base,
           "matcher": "liblgr_hip.so (exact, bit-identical to the oracle)" if a.gpu_matcher else "oracle"}
    if a.by_piece:
        rows = {}
        # every piece ALONE against rounds 1-4's canonical orders (mask 0); then the round-5 default (eigen33 + acosf + atan2f) and all pieces
        # against mask 0; finally what is LEFT between the round-5 default (LGR_ARITH_FAST) and PCL's own arithmetic (LGR_ARITH_PCL)
        dflt = None
        for name, bit in list(o.ARITH_PIECES.items()) + [("round5_default_vs_round4", o.ARITH_CANONICAL), ("all_pieces_vs_round4", o.ARITH_PCL), ("all_pieces_vs_round5_default", o.ARITH_PCL)]:
            other = run_mode(o, pair, bit, matcher, mid)
            if name == "round5_default_vs_round4":
                dflt = other
            c = compare(dflt if name == "all_pieces_vs_round5_default" else can, other, pair)
            rows[name] = {"mode_bits": bit,
                          "normals_differ_in_any_bit": c["normals_src"]["differ_in_any_bit"] + c["normals_tgt"]["differ_in_any_bit"],
                          "normals_max_abs_component_diff": max(c["normals_src"]["max_abs_component_diff"], c["normals_tgt"]["max_abs_component_diff"]),
                          "fpfh_rows_differ_in_any_bit": c["fpfh_src"]["rows_differ_in_any_bit"] + c["fpfh_tgt"]["rows_differ_in_any_bit"],
                          "fpfh_max_abs_bin_diff": max(c["fpfh_src"]["max_abs_bin_diff"], c["fpfh_tgt"]["max_abs_bin_diff"]),
                          "fpfh_rows_with_a_bin_moved_by_more_than_0.5": c["fpfh_src"]["rows_with_a_bin_moved_by_more_than_0.5"] + c["fpfh_tgt"]["rows_with_a_bin_moved_by_more_than_0.5"],
                          "match_indices_differ": c["match"]["src_to_tgt_indices_differ"] + c["match"]["tgt_to_src_indices_differ"],
                          "match_queries": c["match"]["queries"],
                          "correspondences_only_canonical": c["correspondences"]["only_canonical"],
                          "correspondences_only_piece": c["correspondences"]["only_pcl_order"],
                          "correspondences_canonical": c["correspondences"]["canonical"],
                          "inliers": [c["ransac"]["canonical"]["inliers"], c["ransac"]["pcl_order"]["inliers"]],
                          "max_abs_dT": c["ransac"]["max_abs_dT"],
                          "max_abs_err_vs_ground_truth": c["ransac"]["max_abs_err_vs_ground_truth"]}
            if name.startswith("all_pieces"):
                rep[name + "_detail"] = c
            print("# %-30s match indices %7d  correspondences -%d +%d  |dT| %.3e" % (name, rows[name]["match_indices_differ"], rows[name]["correspondences_only_canonical"],
                                                                                   rows[name]["correspondences_only_piece"], rows[name]["max_abs_dT"]), file=sys.stderr, flush=True)
        rep["by_piece"] = rows
    else:
        pcl = run_mode(o, pair, a.mode, matcher, mid)
        rep["pcl_order_pieces"] = {k: bool(a.mode & v) for k, v in o.ARITH_PIECES.items()}
        rep.update(compare(can, pcl, pair))
    rep["seconds"] = time.time() - t0
    line = json.dumps(rep)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
