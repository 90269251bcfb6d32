"""Experiment: P scan pairs in flight on one GPU -- P host threads, each with its own lgr_ctx (own stream +
workspace), each aligning its own stream of pairs; throughput against the serial run, results compared bit for bit.

    python tools/exp_inflight.py [--points 1000000] [--pairs 16] [--inflight 1 2 3] [--single-context]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--pairs", type=int, default=16, help="aligns per configuration")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic pairs cycled through")
    ap.add_argument("--inflight", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--single-context", action="store_true")
    ap.add_argument("--take-turns", action="store_true", help="leave lgr_ctx_options.concurrent_contexts at its default 0")
    ap.add_argument("--matching", default="lr")
    a = ap.parse_args()
    import torch
    from lgr_amd import capi, synthetic
    mid = {"lr": 0, "one_sided": 1, "cluster": 2}[a.matching]
    pairs = [synthetic.make_pair(a.points, seed=synthetic.SEED + i) for i in range(a.distinct)]
    dev = [(torch.from_numpy(p["src"]).cuda(), torch.from_numpy(p["tgt"]).cuda()) for p in pairs]
    params = [capi.default_params(matching_id=mid, metric_id=1, score_id=2, feature_radius=0.25, bf_block_size=200000, max_iterations=1000000,
                                  distance_thr=0.1, vp_src=p["vp_src"], vp_tgt=p["vp_tgt"]) for p in pairs]
    torch.cuda.synchronize()
    ref = {}
    for P in a.inflight:
        ctxs = [capi.Context(0, stream=-1) for _ in range(P)]         # LGR_STREAM_OWN: a non-blocking stream per context
        for c in ctxs:   # (concurrent_contexts = 1: contexts that really overlap on the device; the product default makes them take turns)
            c.set_options(helper_contexts=0 if a.single_context else 1, concurrent_contexts=0 if a.take_turns else 1)
        results = [None] * a.pairs

        def worker(w, lo, hi, out):
            for j in range(lo, hi):
                k = j % a.distinct
                r = ctxs[w].align(dev[k][0], dev[k][1], params[k])
                out[j] = (k, r.matrix().copy(), r.iterations, r.n_inliers, r.n_correspondences)

        for c in range(P):                                              # warm every context (workspace growth, helper threads)
            warm = [None] * a.distinct
            worker(c, 0, a.distinct, warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(w, w * a.pairs // P, (w + 1) * a.pairs // P, results)) for w in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = True
        for r in results:
            k = r[0]
            if k not in ref:
                ref[k] = r
            ok = np.array_equal(ref[k][1].view(np.uint32), r[1].view(np.uint32)) and ref[k][2:] == r[2:]
            if not ok:
                print("   pair %d differs: iterations / inliers / correspondences %s vs %s, max |dT| %.3g" % (k, r[2:], ref[k][2:], np.abs(r[1] - ref[k][1]).max()))
            same = same and ok
        print("in flight %d: %d aligns in %.1f ms -> %.2f ms per pair, %.2f registrations/s, host threads per context %d, identical to the first run: %s"
              % (P, a.pairs, 1e3 * dt, 1e3 * dt / a.pairs, a.pairs / dt, ctxs[0].host_threads(), same), flush=True)
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
