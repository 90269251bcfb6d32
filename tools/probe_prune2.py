"""GPU probe (analysis only): two-pass exact pruning emulation with realistic upper bounds, both match directions.
rows = A sorted by A-leaf, blocks of B rows; columns = B leaves.  Pass 1 visits, for each row block, its T nearest
B leaves (by lower bound) and, for each B leaf, its T nearest row blocks; pass 2 visits every other (block, leaf)
whose lower bound does not exceed the block's or the leaf's largest pass-1 upper bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np, torch
from lgr_amd import capi, synthetic
from probe_prune import features, kmeans, assign


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ctx = capi.Context(0)
    pair = synthetic.make_pair(n, seed=566)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    fa, fb = features(ctx, src, pair["vp_src"]), features(ctx, tgt, pair["vp_tgt"])
    ctx.sync()
    for P, B, T in ((1024, 256, 2), (1024, 256, 4), (2048, 256, 4), (1024, 128, 4)):
        t0 = time.time()
        cb, lab_b = kmeans(fb, P)
        ob = torch.argsort(lab_b)
        fbs, lab_bs = fb[ob], lab_b[ob]
        start_b = torch.searchsorted(lab_bs, torch.arange(P + 1, device="cuda"))
        size_b = (start_b[1:] - start_b[:-1]).float()
        r_b = torch.zeros(P, device="cuda").scatter_reduce_(0, lab_bs, (fbs - cb[lab_bs]).norm(dim=1), "amax")
        ca, lab_a = kmeans(fa, P, seed=1)
        oa = torch.argsort(lab_a)
        fas = fa[oa]
        nblk = fas.shape[0] // B
        fas = fas[: nblk * B]
        # lower bounds LB[block, leaf] = max(0, min_i |a_i - c_leaf| - r_leaf)
        LB = torch.empty(nblk, P, device="cuda")
        for s in range(0, nblk, 256):
            e = min(s + 256, nblk)
            d = torch.cdist(fas[s * B:e * B], cb).reshape(e - s, B, P)
            LB[s:e] = (d.min(dim=1).values - r_b[None, :]).clamp(min=0)
        visit = torch.zeros(nblk, P, dtype=torch.bool, device="cuda")
        visit.scatter_(1, LB.topk(T, dim=1, largest=False).indices, True)
        visit.scatter_(0, LB.topk(T, dim=0, largest=False).indices, True)
        # pass 1: exact distances on visited tiles
        Ua = torch.full((nblk * B,), float("inf"), device="cuda")
        Ub = torch.full((fbs.shape[0],), float("inf"), device="cuda")
        vi = visit.nonzero()
        for blk, leaf in vi.tolist():
            s, e = int(start_b[leaf]), int(start_b[leaf + 1])
            if e == s: continue
            d = torch.cdist(fas[blk * B:(blk + 1) * B], fbs[s:e])
            Ua[blk * B:(blk + 1) * B] = torch.minimum(Ua[blk * B:(blk + 1) * B], d.min(dim=1).values)
            Ub[s:e] = torch.minimum(Ub[s:e], d.min(dim=0).values)
        Ua_blk = Ua.reshape(nblk, B).max(dim=1).values
        Ub_leaf = torch.zeros(P, device="cuda").scatter_reduce_(0, lab_bs, Ub, "amax")
        w1 = (visit.float() * size_b[None, :]).sum().item()
        need_row = LB <= Ua_blk[:, None] * 1.01 + 0.05
        need_col = LB <= Ub_leaf[None, :] * 1.01 + 0.05
        w2_both = (((need_row | need_col) & ~visit).float() * size_b[None, :]).sum().item()
        w2_row = ((need_row & ~visit).float() * size_b[None, :]).sum().item()
        tot = nblk * float(fbs.shape[0])
        print(f"P={P} B={B} T={T}: pass1 {w1 / tot:.4f}  pass2(both dirs) {w2_both / tot:.4f}  pass2(rows only) {w2_row / tot:.4f}  "
              f"total(both) {(w1 + w2_both) / tot:.4f}  items p1 {int(visit.sum())} p2 {int(((need_row | need_col) & ~visit).sum())}  "
              f"Ua_blk med {Ua_blk.median().item():.2f} Ub_leaf med {Ub_leaf.median().item():.2f} inf leaves {int(torch.isinf(Ub_leaf).sum())} [{time.time() - t0:.1f}s]", flush=True)


if __name__ == "__main__":
    main()
