"""GPU probe (analysis only): do slab bounds along each leaf's principal axes tighten the ball bound enough to matter?
LB(rb, g) = max(ball bound, max_k slab_k bound), slab_k: projections of the leaf's rows on its k-th principal axis span
[lo, hi]; a row block whose rows project to [pmin, pmax] is at least max(0, pmin - hi, lo - pmax) away."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi, synthetic
from probe_prune import features
from probe_prune3 import two_level, balls


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ctx = capi.Context(0)
    pair = synthetic.make_pair(n, seed=566)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    fa, fb = features(ctx, src, pair["vp_src"]), features(ctx, tgt, pair["vp_tgt"])
    ctx.sync()
    RB, T = 256, 64
    NPC = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    K2 = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    la = two_level(fa, 16, K2, 1); lb = two_level(fb, 16, K2, 2)
    oka, okb = torch.isfinite(fa).all(1), torch.isfinite(fb).all(1)
    fa, la, fb, lb = fa[oka], la[oka], fb[okb], lb[okb]
    mode = sys.argv[3] if len(sys.argv) > 3 else "leaf"
    if mode == "pc1":        # rows inside a leaf ordered along the first global principal axis: row blocks become slabs
        mu0 = fb.mean(0, keepdim=True)
        _, _, V0 = torch.pca_lowrank(fb - mu0, q=2, center=False)
        key = ((fa - mu0) @ V0[:, 0])
        oa = torch.argsort(key); oa = oa[torch.argsort(la[oa], stable=True)]
    elif mode == "radial":   # rows inside a leaf ordered by the distance to the global mean (like the cluster-centre order in use)
        key = (fa - fa.mean(0, keepdim=True)).norm(dim=1)
        oa = torch.argsort(key); oa = oa[torch.argsort(la[oa], stable=True)]
    else:
        oa = torch.argsort(la, stable=True)
    ob = torch.argsort(lb, stable=True)
    fas, fbs, lbs = fa[oa], fb[ob], lb[ob]
    nblk = fas.shape[0] // RB
    fas = fas[: nblk * RB]
    # column groups = the leaves themselves (variable sizes)
    leaves, counts = torch.unique_consecutive(lbs, return_counts=True)
    ncg = leaves.numel()
    starts = torch.cumsum(counts, 0) - counts
    gid = torch.repeat_interleave(torch.arange(ncg, device="cuda"), counts)
    mb = torch.zeros(ncg, 33, device="cuda").index_add_(0, gid, fbs) / counts[:, None]
    dev = fbs - mb[gid]
    rb_ = torch.zeros(ncg, device="cuda").scatter_reduce_(0, gid, dev.norm(dim=1), "amax")
    cov = torch.zeros(ncg, 33, 33, device="cuda").index_add_(0, gid, dev[:, :, None] * dev[:, None, :]) / counts[:, None, None]
    evals, evecs = torch.linalg.eigh(cov)
    U = evecs[:, :, -NPC:].flip(-1)                        # [ncg, 33, NPC]
    proj = torch.einsum("ij,ijk->ik", dev, U[gid])         # [cols, NPC]
    lo = torch.full((ncg, NPC), float("inf"), device="cuda").scatter_reduce_(0, gid[:, None].expand(-1, NPC), proj, "amin")
    hi = torch.full((ncg, NPC), float("-inf"), device="cuda").scatter_reduce_(0, gid[:, None].expand(-1, NPC), proj, "amax")
    print("leaves %d, radius med %.1f; half-widths of the first %d principal slabs med: %s" % (ncg, rb_.median().item(), NPC, [round(v, 1) for v in ((hi - lo) / 2).median(0).values.tolist()]), flush=True)
    LB_ball = torch.empty(nblk, ncg, device="cuda"); LB_slab = torch.zeros(nblk, ncg, device="cuda")
    mU = torch.einsum("gj,gjk->gk", mb, U)
    for s in range(0, nblk, 32):
        e = min(s + 32, nblk)
        X = fas[s * RB:e * RB]
        d = torch.cdist(X, mb).reshape(e - s, RB, ncg)
        LB_ball[s:e] = (d.min(1).values - rb_[None, :]).clamp(min=0)
        P = (torch.einsum("ij,gjk->igk", X, U) - mU[None]).reshape(e - s, RB, ncg, NPC)
        pmin, pmax = P.min(1).values, P.max(1).values
        LB_slab[s:e] = torch.maximum(pmin - hi[None], lo[None] - pmax).clamp(min=0).max(-1).values
    def box_lb(A, B):
        # bounding boxes of row blocks / leaves in the coordinates given; LB^2 = sum over axes of the squared gaps
        amin = A.reshape(nblk, RB, -1).min(1).values; amax = A.reshape(nblk, RB, -1).max(1).values
        K = A.shape[1]
        blo = torch.full((ncg, K), float("inf"), device="cuda").scatter_reduce_(0, gid[:, None].expand(-1, K), B, "amin")
        bhi = torch.full((ncg, K), float("-inf"), device="cuda").scatter_reduce_(0, gid[:, None].expand(-1, K), B, "amax")
        out = torch.empty(nblk, ncg, device="cuda")
        for s0 in range(0, nblk, 256):
            e0 = min(s0 + 256, nblk)
            gap = torch.maximum(amin[s0:e0, None, :] - bhi[None], blo[None] - amax[s0:e0, None, :]).clamp(min=0)
            out[s0:e0] = gap.pow(2).sum(-1).sqrt()
        return out
    LB_box = box_lb(fas, fbs)
    mu = fbs.mean(0, keepdim=True)
    _, _, Vg = torch.pca_lowrank(fbs - mu, q=33, center=False)
    LB_boxp = box_lb((fas - mu) @ Vg, (fbs - mu) @ Vg)
    w = counts.float() / counts.float().sum()              # tile fractions are weighted by the leaf sizes
    CGs = counts
    for name, LB in (("ball", LB_ball), ("max(ball, slabs)", torch.maximum(LB_ball, LB_slab)), ("max(ball, box raw)", torch.maximum(LB_ball, LB_box)),
                     ("max(ball, box global PCA)", torch.maximum(LB_ball, LB_boxp)), ("max(ball, slabs, box global PCA)", torch.maximum(torch.maximum(LB_ball, LB_slab), LB_boxp))):
        visit = torch.zeros(nblk, ncg, dtype=torch.bool, device="cuda")
        visit.scatter_(1, LB.topk(T, dim=1, largest=False).indices, True)
        visit.scatter_(0, LB.topk(T, dim=0, largest=False).indices, True)
        Ua = torch.full((nblk * RB,), float("inf"), device="cuda"); Ub = torch.full((fbs.shape[0],), float("inf"), device="cuda")
        vis = visit.nonzero()
        for g in vis[:, 1].unique().tolist():
            blks = vis[vis[:, 1] == g, 0]
            rows = (blks[:, None] * RB + torch.arange(RB, device="cuda")[None, :]).reshape(-1)
            c0, c1 = int(starts[g]), int(starts[g] + counts[g])
            d = torch.cdist(fas[rows], fbs[c0:c1])
            Ua[rows] = torch.minimum(Ua[rows], d.min(dim=1).values)
            Ub[c0:c1] = torch.minimum(Ub[c0:c1], d.min(dim=0).values)
        Ua_blk = Ua.reshape(nblk, RB).max(dim=1).values
        Ub_g = torch.zeros(ncg, device="cuda").scatter_reduce_(0, gid, Ub, "amax")
        need = ((LB <= Ua_blk[:, None] * 1.01 + 0.05) | (LB <= Ub_g[None, :] * 1.01 + 0.05)) & ~visit
        f0 = (visit.float() * w[None]).sum(1).mean().item(); f1 = (need.float() * w[None]).sum(1).mean().item()
        print(f"{name}: pass0 {f0:.4f} final {f1:.4f} total {f0 + f1:.4f}", flush=True)


if __name__ == "__main__":
    main()
