"""GPU probe (analysis only): tile fraction of the final matcher pass under a per-row / per-column schedule criterion
(tile needed iff ANY of its rows has  |a_i - c_g| - r_g <= U_i,  or any of its columns the mirrored test against the
row block's ball) versus the per-block criterion in use (min_i |a_i - c_g| - r_g <= max_i U_i)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi, synthetic
from probe_prune import features
from probe_prune3 import two_level, balls


def dist_any(xs, RB, m, r, U):
    """returns (LBmin[nblk, ng] = min_i |x_i - m_g| - r_g,  any[nblk, ng] = exists i: |x_i - m_g| - r_g <= U_i)"""
    nblk = xs.shape[0] // RB
    LB = torch.empty(nblk, m.shape[0], device="cuda")
    ANY = torch.empty(nblk, m.shape[0], dtype=torch.bool, device="cuda")
    for s in range(0, nblk, 64):
        e = min(s + 64, nblk)
        d = torch.cdist(xs[s * RB:e * RB], m) - r[None, :]
        LB[s:e] = d.reshape(e - s, RB, -1).min(dim=1).values.clamp(min=0)
        if U is not None:
            ANY[s:e] = (d <= U[s * RB:e * RB, None] * 1.01 + 0.05).reshape(e - s, RB, -1).any(dim=1)
    return LB, ANY


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ctx = capi.Context(0)
    pair = synthetic.make_pair(n, seed=566)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    fa, fb = features(ctx, src, pair["vp_src"]), features(ctx, tgt, pair["vp_tgt"])
    ctx.sync()
    RB = 256
    for k2, CG, T in ((64, 1024, 16), (64, 1024, 64)):
        t0 = time.time()
        fas = fa[torch.argsort(two_level(fa, 16, k2, 1), stable=True)]
        fbs = fb[torch.argsort(two_level(fb, 16, k2, 2), stable=True)]
        fas = fas[torch.isfinite(fas).all(1)]; fbs = fbs[torch.isfinite(fbs).all(1)]
        nblk, ncg = fas.shape[0] // RB, fbs.shape[0] // CG
        fas, fbs = fas[: nblk * RB], fbs[: ncg * CG]
        mb, rb_ = balls(fbs, CG)          # column groups as balls
        ma, ra_ = balls(fas, RB)          # row blocks as balls
        LBr, _ = dist_any(fas, RB, mb, rb_, None)
        LBc, _ = dist_any(fbs, CG, ma, ra_, None)
        LB = torch.maximum(LBr, LBc.T)
        visit = torch.zeros(nblk, ncg, dtype=torch.bool, device="cuda")
        visit.scatter_(1, LB.topk(T, dim=1, largest=False).indices, True)
        visit.scatter_(0, LB.topk(max(1, T * RB // CG), dim=0, largest=False).indices, True)
        Ua = torch.full((nblk * RB,), float("inf"), device="cuda")
        Ub = torch.full((ncg * CG,), float("inf"), device="cuda")
        vis = visit.nonzero()
        for g in vis[:, 1].unique().tolist():      # per column group: all its visited row blocks at once
            blks = vis[vis[:, 1] == g, 0]
            rows = (blks[:, None] * RB + torch.arange(RB, device="cuda")[None, :]).reshape(-1)
            d = torch.cdist(fas[rows], fbs[g * CG:(g + 1) * CG])
            Ua[rows] = torch.minimum(Ua[rows], d.min(dim=1).values)
            Ub[g * CG:(g + 1) * CG] = torch.minimum(Ub[g * CG:(g + 1) * CG], d.min(dim=0).values)
        Ua_blk, Ub_g = Ua.reshape(nblk, RB).max(dim=1).values, Ub.reshape(ncg, CG).max(dim=1).values
        cur = ((LB <= Ua_blk[:, None] * 1.01 + 0.05) | (LB <= Ub_g[None, :] * 1.01 + 0.05)) & ~visit
        _, anyr = dist_any(fas, RB, mb, rb_, Ua)
        _, anyc = dist_any(fbs, CG, ma, ra_, Ub)
        new = (anyr | anyc.T) & ~visit
        half = (anyr | (LB <= Ub_g[None, :] * 1.01 + 0.05)) & ~visit
        # outlier split (rows only): rows with U above the q-quantile leave their block and are re-blocked among themselves
        for qq in (0.8, 0.9, 0.95):
            thr = torch.quantile(Ua[:: max(1, Ua.numel() // 500000)], qq)
            Ub_ = Ua.clone(); Ub_[Ua > thr] = -1e30            # outliers never trigger in their home block
            _, any_bulk = dist_any(fas, RB, mb, rb_, Ub_)
            out_idx = (Ua > thr).nonzero().squeeze(1)
            no = out_idx.numel() // RB * RB
            _, any_out = dist_any(fas[out_idx[:no]], RB, mb, rb_, Ua[out_idx[:no]])
            f_bulk = (any_bulk & ~visit).float().mean().item()
            f_out = any_out.float().mean().item() * (no // RB) / nblk
            print(f"    rows-only outlier split q={qq}: thr {thr.item():.1f}  bulk {f_bulk:.4f} + outliers {f_out:.4f} = {f_bulk + f_out:.4f}", flush=True)
        q = torch.tensor([0.5, 0.9, 0.99], device="cuda")
        print(f"k2={k2} CG={CG} T={T}: pass0 {visit.float().mean().item():.4f}  final: per-block {cur.float().mean().item():.4f}  "
              f"per-row+per-col {new.float().mean().item():.4f} (rows only {(anyr & ~visit).float().mean().item():.4f}, cols only "
              f"{(anyc.T & ~visit).float().mean().item():.4f})  per-row + block-col {half.float().mean().item():.4f}   "
              f"U rows q50/90/99 {[round(v, 1) for v in torch.quantile(Ua[:500000], q).tolist()]} blkmax med {Ua_blk.median().item():.1f} [{time.time() - t0:.1f}s]", flush=True)


if __name__ == "__main__":
    main()
