"""Experiment: how much could a RADIAL lower bound  (| |a - c| - |b - c| |)^2 <= d2(a, b)  prune, c = centre of the row's cluster?
Rows and columns are already sorted by that radius inside their leaves, so (row block, stage) shells are narrow.  For sampled rows
with their true NN distance: the share of columns whose radius about the row's centre lies within sqrt(d_nn) of the row's.
    python tools/exp_radial_bound.py [--points 1000000]
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    a = ap.parse_args()
    import torch
    from lgr_amd import capi, synthetic
    pair = synthetic.make_pair(a.points, seed=synthetic.SEED)
    ctx = capi.Context(0)
    r = 0.25
    voxel = float(np.sqrt(np.float32(np.pi * r * r / 352.0)))
    feats = []
    for side in ("src", "tgt"):
        cloud = torch.from_numpy(pair[side]).cuda()
        surf = ctx.downsample(cloud, voxel); ctx.sync()
        nrm = surf.clone(); torch.cuda.synchronize()
        ctx.normals_knn(nrm, 30, vp=pair["vp_" + side]); ctx.sync()
        f = ctx.fpfh(nrm, nrm, r); ctx.sync()
        feats.append(f.clone())
    A, B = feats
    ok = torch.isfinite(A).all(1); A = A[ok]
    ok = torch.isfinite(B).all(1); B = B[ok]
    print("rows", A.shape, "cols", B.shape)
    m = ctx.match_bf2(A, B, 200000); ctx.sync()
    d_nn = m[1].clone()
    # 16 centres: Lloyd on a sample of both sides
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    S = torch.cat([A[torch.randint(0, A.shape[0], (60000,), device="cuda", generator=g)], B[torch.randint(0, B.shape[0], (60000,), device="cuda", generator=g)]])
    C = S[torch.randperm(S.shape[0], device="cuda", generator=g)[:16]].clone()
    for it in range(15):
        lab = torch.cdist(S, C).argmin(1)
        for c in range(16):
            sel = lab == c
            if sel.any(): C[c] = S[sel].mean(0)
    dA = torch.cdist(A, C); labA = dA.argmin(1); rA = dA.gather(1, labA[:, None])[:, 0]
    dB = torch.cdist(B, C); labB = dB.argmin(1)
    idx = torch.randint(0, A.shape[0], (4000,), device="cuda", generator=g)
    f_all, f_same, n_same = [], [], []
    for i in idx.tolist():
        c = int(labA[i]); ra = rA[i]; dn = torch.sqrt(d_nn[i]) * 1.0
        rb = dB[:, c]
        near = (rb - ra).abs() <= dn
        same = labB == c
        f_all.append(float(near.float().mean()))
        f_same.append(float((near & same).sum()) / max(1.0, float(same.sum())))
        n_same.append(float(same.float().mean()))
    f_all, f_same, n_same = np.array(f_all), np.array(f_same), np.array(n_same)
    print("NN distance sqrt: median %.3f; row radius median %.3f" % (float(torch.sqrt(d_nn[idx]).median()), float(rA[idx].median())))
    print("share of ALL columns within the radial band: mean %.4f median %.4f" % (f_all.mean(), np.median(f_all)))
    print("share of SAME-cluster columns within the band: mean %.4f median %.4f (same-cluster share of all columns: %.4f)" % (f_same.mean(), np.median(f_same), n_same.mean()))
    # block-level version: rows sorted by (cluster, radius) in blocks of 256, thresholds = max d_nn of the block
    order = torch.argsort(labA.double() * 1e6 + rA.double())
    blk = order[: (order.shape[0] // 256) * 256].view(-1, 256)
    bsel = torch.randint(0, blk.shape[0], (600,), device="cuda", generator=g)
    fb = []
    for b in bsel.tolist():
        rows = blk[b]
        c = int(labA[rows[0]])
        if int(labA[rows[-1]]) != c: continue
        lo, hi = rA[rows].min(), rA[rows].max()
        dn = torch.sqrt(d_nn[rows].max())
        rb = dB[:, c]
        near = (rb >= lo - dn) & (rb <= hi + dn)
        same = labB == c
        fb.append((float(near.float().mean()), float((near & same).sum()) / max(1.0, float(same.sum())), float(hi - lo), float(dn)))
    fb = np.array(fb)
    print("256-row blocks (cluster, radius order): share of all columns in the band %.4f, of same-cluster columns %.4f; shell width %.3f, sqrt(max d_nn) %.3f" % tuple(fb.mean(0)))

if __name__ == "__main__":
    main()
