"""GPU probe (analysis only): the skip scheme the kernel can implement -- rows/cols sorted by a two-level k-means
label, fixed row blocks (RB rows) and column groups (CG cols) with bounding balls, two passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi, synthetic
from probe_prune import features, kmeans, assign


def two_level(x, k1, k2, seed):
    c1, l1 = kmeans(x, k1, iters=6, seed=seed)
    lab = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
    for p in range(k1):
        idx = (l1 == p).nonzero().squeeze(1)
        if idx.numel() == 0: continue
        k = min(k2, idx.numel())
        c2, l2 = kmeans(x[idx], k, iters=6, seed=seed + p)
        # order the sub-leaves along the first principal direction of their centres (keeps neighbours adjacent)
        cc = c2 - c2.mean(0, keepdim=True)
        _, _, v = torch.pca_lowrank(cc, q=1)
        rank = torch.argsort(torch.argsort((cc @ v[:, 0])))
        lab[idx] = p * k2 + rank[l2]
    return lab


def balls(xs, G):
    n = xs.shape[0] // G
    g = xs[: n * G].reshape(n, G, 33)
    m = g.mean(dim=1)
    r = (g - m[:, None, :]).norm(dim=2).max(dim=1).values
    return m, r


def point_ball_lb(xs, RB, m, r):
    nblk = xs.shape[0] // RB
    LB = torch.empty(nblk, m.shape[0], device="cuda")
    for s in range(0, nblk, 128):
        e = min(s + 128, nblk)
        d = torch.cdist(xs[s * RB:e * RB], m).reshape(e - s, RB, -1)
        LB[s:e] = (d.min(dim=1).values - r[None, :]).clamp(min=0)
    return LB


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ctx = capi.Context(0)
    pair = synthetic.make_pair(n, seed=566)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    fa, fb = features(ctx, src, pair["vp_src"]), features(ctx, tgt, pair["vp_tgt"])
    ctx.sync()
    for k2, RB, CG, T in ((64, 256, 512, 4), (64, 256, 256, 4), (256, 256, 512, 4), (64, 256, 512, 8)):
        t0 = time.time()
        fas = fa[torch.argsort(two_level(fa, 16, k2, 1), stable=True)]
        fbs = fb[torch.argsort(two_level(fb, 16, k2, 2), stable=True)]
        nblk, ncg = fas.shape[0] // RB, fbs.shape[0] // CG
        fas, fbs = fas[: nblk * RB], fbs[: ncg * CG]
        mb, rb_ = balls(fbs, CG)
        ma, ra_ = balls(fas, RB)
        LB = torch.maximum(point_ball_lb(fas, RB, mb, rb_), point_ball_lb(fbs, CG, ma, ra_).T)
        visit = torch.zeros(nblk, ncg, dtype=torch.bool, device="cuda")
        visit.scatter_(1, LB.topk(T, dim=1, largest=False).indices, True)
        visit.scatter_(0, LB.topk(T, dim=0, largest=False).indices, True)
        Ua = torch.full((nblk * RB,), float("inf"), device="cuda")
        Ub = torch.full((ncg * CG,), float("inf"), device="cuda")
        for blk, g in visit.nonzero().tolist():
            d = torch.cdist(fas[blk * RB:(blk + 1) * RB], fbs[g * CG:(g + 1) * CG])
            Ua[blk * RB:(blk + 1) * RB] = torch.minimum(Ua[blk * RB:(blk + 1) * RB], d.min(dim=1).values)
            Ub[g * CG:(g + 1) * CG] = torch.minimum(Ub[g * CG:(g + 1) * CG], d.min(dim=0).values)
        Ua_blk, Ub_g = Ua.reshape(nblk, RB).max(dim=1).values, Ub.reshape(ncg, CG).max(dim=1).values
        need = ((LB <= Ua_blk[:, None] * 1.01 + 0.05) | (LB <= Ub_g[None, :] * 1.01 + 0.05)) & ~visit
        need_row = (LB <= Ua_blk[:, None] * 1.01 + 0.05) & ~visit
        w1, w2, w2r = visit.float().mean().item(), need.float().mean().item(), need_row.float().mean().item()
        print(f"k2={k2} RB={RB} CG={CG} T={T}: pass1 {w1:.4f} pass2 {w2:.4f} (rows only {w2r:.4f}) total {w1 + w2:.4f}  "
              f"ball r med: cols {rb_.median().item():.1f} rows {ra_.median().item():.1f}  Ua_blk med {Ua_blk.median().item():.1f} "
              f"Ub_g med {Ub_g.median().item():.1f} [{time.time() - t0:.1f}s]", flush=True)


if __name__ == "__main__":
    main()
