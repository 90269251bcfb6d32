"""Diagnostic: the stand-alone k-NN normals stage looped on one context while another host thread keeps the GPU busy with a chosen
load on a second context; every normals result compared bit for bit with the serial one.

    python tools/exp_normals_contention.py --load align|normals|match|fpfh|none [--rounds 300]
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
    ap.add_argument("--load", default="align")
    ap.add_argument("--rounds", type=int, default=300)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--stage", default="normals", choices=["normals", "fpfh", "downsample", "knn"])
    a = ap.parse_args()
    import torch
    from lgr_amd import capi, synthetic
    pair = synthetic.make_pair(a.points, seed=synthetic.SEED)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    params = capi.default_params(matching_id=0, metric_id=1, score_id=2, feature_radius=0.25, bf_block_size=200000, max_iterations=1000000,
                                 distance_thr=0.1, vp_src=pair["vp_src"], vp_tgt=pair["vp_tgt"])
    voxel = float(np.sqrt(np.float32(np.pi * 0.25 * 0.25 / 352.0)))
    A, B = capi.Context(0, stream=-1), capi.Context(0, stream=-1)
    surf = A.downsample(src, voxel); A.sync(); surf = surf.clone(); torch.cuda.synchronize()
    nrm0 = surf.clone(); torch.cuda.synchronize(); A.normals_knn(nrm0, 30, vp=pair["vp_src"]); A.sync()
    fB = None
    if a.load in ("match", "fpfh"):
        fs = B.fpfh(src, nrm0, 0.25); B.sync()
        nt = B.downsample(tgt, voxel); B.sync(); nt = nt.clone(); torch.cuda.synchronize(); B.normals_knn(nt, 30, vp=pair["vp_tgt"]); B.sync()
        ft = B.fpfh(tgt, nt, 0.25); B.sync()
        fB = (fs, ft)

    def stage():
        if a.stage == "normals":
            x = surf.clone(); torch.cuda.synchronize(); A.normals_knn(x, 30, vp=pair["vp_src"]); A.sync(); return x.cpu().numpy()
        if a.stage == "fpfh":
            x = A.fpfh(src, nrm0, 0.25); A.sync(); return x.cpu().numpy()
        if a.stage == "downsample":
            x = A.downsample(src, voxel); A.sync(); return x.cpu().numpy()
        i, d = A.knn(surf, surf, 30); A.sync(); return np.concatenate([i.cpu().numpy().view(np.uint32), d.cpu().numpy().view(np.uint32)], 1)

    ref = stage()
    assert np.array_equal(ref.view(np.uint32), stage().view(np.uint32))
    stop = [False]

    def load():
        while not stop[0]:
            if a.load == "align":
                B.align(src, tgt, params)
            elif a.load == "normals":
                y = surf.clone(); torch.cuda.synchronize(); B.normals_knn(y, 30, vp=pair["vp_src"]); B.sync()
            elif a.load == "match":
                B.match_bf2(fB[0], fB[1], 200000); B.sync()
            elif a.load == "fpfh":
                B.fpfh(src, nrm0, 0.25); B.sync()
            else:
                time.sleep(0.01)

    th = threading.Thread(target=load)
    th.start()
    bad = 0
    t0 = time.time()
    for r in range(a.rounds):
        got = stage()
        d = np.flatnonzero(ref.reshape(-1).view(np.uint32) != got.reshape(-1).view(np.uint32))
        if len(d):
            bad += 1
            if bad <= 5:
                w = ref.shape[1]
                rows = np.unique(d // w)
                print("round %d (%.1f s): %d words differ in %d rows, columns %s, rows %s" % (r, time.time() - t0, len(d), len(rows), np.unique(d % w).tolist()[:12], rows[:10].tolist()), flush=True)
    stop[0] = True
    th.join()
    print("stage %s under load %s: %d of %d rounds differ" % (a.stage, a.load, bad, a.rounds), flush=True)


if __name__ == "__main__":
    main()
