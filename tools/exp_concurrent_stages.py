"""Diagnostic: the staged pipeline (downsample -> normals -> FPFH -> match both ways -> filter -> RANSAC) run concurrently from P
host threads on P contexts (own streams), every stage output compared bit for bit with a serial reference run of the same pair.
Localises any cross-context interference to the first stage whose output differs.

    python tools/exp_concurrent_stages.py [--points 1000000] [--threads 2] [--rounds 6]
"""
import argparse
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))


def stages(ctx, capi, torch, pair, src, tgt, params):
    r = 0.25
    voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))
    out = {}
    feats = []
    for side, cloud in (("src", src), ("tgt", tgt)):
        # (the contexts run on streams of their own here: torch's copies on ITS stream must wait for the library's stream by hand)
        surf = ctx.downsample(cloud, voxel); ctx.sync(); surf = surf.clone(); nrm = surf.clone(); torch.cuda.synchronize()
        ctx.normals_knn(nrm, 30, vp=pair["vp_" + side]); ctx.sync()
        feat = ctx.fpfh(cloud, nrm, r)
        ctx.sync()
        out["surf_" + side] = surf.cpu().numpy(); out["nrm_" + side] = nrm.cpu().numpy(); out["feat_" + side] = feat.cpu().numpy()
        feats.append(feat)
    m = ctx.match_bf2(feats[0], feats[1], 200000)
    ctx.sync()
    for n, x in zip(("ab_i", "ab_d", "ba_i", "ba_d"), m):
        out[n] = x.cpu().numpy()
    corr = ctx.filter(params.matching_id, src, tgt, *m, params.distance_thr)      # (syncs: the count is read back; .cpu() inside waits on torch's stream only)
    out["corr"] = corr.view(np.uint32).reshape(-1, 4).copy()
    res, mask = ctx.ransac(src, tgt, corr, params)
    out["T"] = res.matrix(); out["mask"] = mask
    out["ransac"] = np.array([res.iterations, res.n_inliers, res.best_iteration, res.num_rejections], np.int64)
    full = ctx.align(src, tgt, params)
    out["align_T"] = full.matrix()
    out["align"] = np.array([full.iterations, full.n_inliers, full.n_correspondences], np.int64)
    return out


ORDER = ["surf_src", "nrm_src", "feat_src", "surf_tgt", "nrm_tgt", "feat_tgt", "ab_i", "ab_d", "ba_i", "ba_d", "corr", "ransac", "mask", "T", "align", "align_T"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--matching", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=2, help="distinct pairs; thread w starts at pair w and alternates")
    ap.add_argument("--single-context", action="store_true")
    ap.add_argument("--take-turns", action="store_true", help="leave lgr_ctx_options.concurrent_contexts at its default 0 (the product default)")
    ap.add_argument("--align-only", action="store_true", help="only the one-call pipeline (no stand-alone stages, no syncs in between)")
    a = ap.parse_args()
    import torch
    from lgr_amd import capi, synthetic
    pairs = [synthetic.make_pair(a.points, seed=synthetic.SEED + i) for i in range(a.distinct)]
    dev = [(torch.from_numpy(p["src"]).cuda(), torch.from_numpy(p["tgt"]).cuda()) for p in pairs]
    prm = [capi.default_params(matching_id=a.matching, metric_id=1, score_id=2, feature_radius=0.25, bf_block_size=200000, max_iterations=1000000,
                               distance_thr=0.1, vp_src=p["vp_src"], vp_tgt=p["vp_tgt"]) for p in pairs]
    torch.cuda.synchronize()

    def run(ctx, k):
        if a.align_only:
            import ctypes as C
            full = ctx.align(dev[k][0], dev[k][1], prm[k])
            out = {}
            n = dev[k][0].shape[0]
            for name, count, dt in (("surf_s", 0, np.float32), ("surf_t", 0, np.float32), ("feat_s", n * 33, np.float32), ("feat_t", n * 33, np.float32),
                                    ("thr", 2 * n, np.float32), ("ij", n, np.int32), ("dij", n, np.float32), ("ji", n, np.int32), ("dji", n, np.float32),
                                    ("corr", full.n_correspondences * 4, np.int32)):
                ptr, cap = C.c_void_p(), C.c_size_t()
                assert capi.lib().lgr_debug_ws(ctx.h, name.encode(), C.byref(ptr), C.byref(cap)) == 0
                if count == 0:
                    count = 600000 * 12          # the surface clouds: a prefix (their sizes are not exported)
                host = np.empty(count, dt)
                assert hip.hipMemcpy(C.c_void_p(host.ctypes.data), ptr, C.c_size_t(count * 4), 2) == 0
                out["ws_" + name] = host
            out["align"] = np.array([full.iterations, full.n_inliers, full.n_correspondences], np.int64)
            out["align_T"] = full.matrix()
            return out
        return stages(ctx, capi, torch, pairs[k], dev[k][0], dev[k][1], prm[k])

    order = ["ws_surf_s", "ws_surf_t", "ws_feat_s", "ws_feat_t", "ws_thr", "ws_ij", "ws_dij", "ws_ji", "ws_dji", "ws_corr", "align", "align_T"] if a.align_only else ORDER
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    c0 = capi.Context(0, stream=-1)
    refs = [run(c0, k) for k in range(a.distinct)]
    for k in range(a.distinct):
        again = run(c0, k)
        for n in order:
            if not np.array_equal(refs[k][n].view(np.uint8), again[n].view(np.uint8)):
                print("SERIAL rerun of pair %d differs at %s" % (k, n), flush=True)
    c0.close()
    bad = []
    lock = threading.Lock()

    def worker(w):
        ctx = capi.Context(0, stream=-1)
        # (concurrent_contexts = 1: the experiment is about contexts that really overlap on the device; by default they take turns)
        ctx.set_options(helper_contexts=0 if a.single_context else 1, concurrent_contexts=0 if a.take_turns else 1)
        for it in range(a.rounds):
            k = (w + it) % a.distinct
            got = run(ctx, k)
            for n in order:
                x, y = refs[k][n], got[n]
                if x.shape != y.shape or not np.array_equal(x.view(np.uint8), y.view(np.uint8)):
                    nb = int((x.reshape(-1).view(np.uint8) != y.reshape(-1).view(np.uint8)).sum()) if x.shape == y.shape else -1
                    with lock:
                        bad.append((w, it, n, nb))
                    print("thread %d round %d pair %d: first difference at %s (%d differing bytes, shapes %s %s) %s" % (w, it, k, n, nb, x.shape, y.shape,
                          (x, y) if x.size <= 4 else ""), flush=True)
                    if x.shape == y.shape and x.size > 4:
                        d = np.flatnonzero(x.reshape(-1).view(np.uint32) != y.reshape(-1).view(np.uint32))
                        per = 12 if "surf" in n else (33 if "feat" in n else (4 if "corr" in n else 1))
                        rows = np.unique(d // per)
                        print("   %d differing words in %d rows (row stride %d); columns hit %s; rows %s ... %s; first values ref %s got %s" % (
                            len(d), len(rows), per, np.unique(d % per).tolist(), rows[:12].tolist(), rows[-4:].tolist(),
                            x.reshape(-1)[d[:6]].tolist(), y.reshape(-1)[d[:6]].tolist()), flush=True)
                        if "surf" in n:
                            xu, yu = x.reshape(-1).view(np.int32)[d].astype(np.int64), y.reshape(-1).view(np.int32)[d].astype(np.int64)
                            ul = np.abs(xu - yu)
                            print("   ulp distance of the differing words: max %d, histogram 1:%d 2:%d 3-8:%d >8:%d" % (ul.max(), (ul == 1).sum(), (ul == 2).sum(), ((ul > 2) & (ul <= 8)).sum(), (ul > 8).sum()), flush=True)
                        gaps = np.diff(rows)
                        print("   row gaps: min %d median %d max %d; runs of consecutive rows: %d" % (gaps.min() if len(gaps) else 0, int(np.median(gaps)) if len(gaps) else 0,
                                                                                                  gaps.max() if len(gaps) else 0, int((gaps > 1).sum()) + 1), flush=True)
                    break
        ctx.close()

    th = [threading.Thread(target=worker, args=(w,)) for w in range(a.threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    from lgr_amd import diagnostics
    print("concurrent runs: %d threads x %d rounds, %d with a difference (GPU serial %s, library %s)" % (a.threads, a.rounds, len(bad), diagnostics.gpu_serial(),
                                                                                                         os.environ.get("LGR_HIP_LIB", "in-tree")), flush=True)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
