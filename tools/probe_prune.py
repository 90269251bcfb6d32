"""GPU probe (analysis only): how much of the M x M descriptor-distance work could an exact, bound-based skip remove?

Builds the FPFH rows of the bench pair through the C ABI, clusters the train rows into P leaves (torch k-means),
and evaluates for query blocks of B rows (queries sorted by leaf) x train leaves the fraction of (block, leaf) work
whose lower bound  min_i(|q_i - c_leaf|) - r_leaf  does not exceed the block's largest true NN distance.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np
import torch
from lgr_amd import capi, synthetic


def features(ctx, cloud, vp, radius=0.25, nr=352):
    voxel = float(np.sqrt(np.pi * radius * radius / nr))
    surf = ctx.downsample(cloud, voxel).clone()
    ctx.normals_knn(surf, 30, None, vp)
    return ctx.fpfh(cloud, surf, radius)


def kmeans(x, P, iters=8, seed=0):
    g = torch.Generator(device=x.device).manual_seed(seed)
    c = x[torch.randperm(x.shape[0], device=x.device, generator=g)[:P]].clone()
    for _ in range(iters):
        lab = assign(x, c)
        cnt = torch.bincount(lab, minlength=P).clamp(min=1).float()
        c = torch.zeros_like(c).index_add_(0, lab, x) / cnt[:, None]
    return c, assign(x, c)


def assign(x, c, chunk=65536):
    out = torch.empty(x.shape[0], dtype=torch.long, device=x.device)
    cn = (c * c).sum(1)
    for s in range(0, x.shape[0], chunk):
        d = cn[None, :] - 2 * x[s:s + chunk] @ c.T
        out[s:s + chunk] = d.argmin(1)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ctx = capi.Context(0)
    pair = synthetic.make_pair(n, seed=566)
    src, tgt = torch.from_numpy(pair["src"]).cuda(), torch.from_numpy(pair["tgt"]).cuda()
    fa, fb = features(ctx, src, pair["vp_src"]), features(ctx, tgt, pair["vp_tgt"])
    ctx.sync()
    ab_i, ab_d, ba_i, ba_d = ctx.match_bf2(fa, fb, 200000)
    ctx.sync()
    nn = ab_d.clone()
    print("NN dist quantiles", torch.quantile(nn[:200000], torch.tensor([0.1, 0.5, 0.9, 0.99], device="cuda")).tolist(), flush=True)
    print("norm quantiles", torch.quantile(fa.norm(dim=1)[:200000], torch.tensor([0.1, 0.5, 0.9], device="cuda")).tolist(), flush=True)
    for P in (256, 1024, 4096):
        t = time.time()
        c, lab_b = kmeans(fb, P)
        # leaf radius and size
        db = (fb - c[lab_b]).norm(dim=1)
        r = torch.zeros(P, device="cuda").scatter_reduce_(0, lab_b, db, "amax")
        size = torch.bincount(lab_b, minlength=P).float()
        lab_a = assign(fa, c)
        order = torch.argsort(lab_a * 1000.0 + (fa - c[lab_a]).norm(dim=1) / 200.0)   # by leaf then radius
        qa, nn_s = fa[order], nn[order]
        for B in (256, 1024):
            nblk = qa.shape[0] // B
            work = 0.0
            work_slack = 0.0
            for s in range(0, nblk, 256):
                blk = qa[s * B:min(s + 256, nblk) * B].reshape(-1, B, 33)
                nb = blk.shape[0]
                d = torch.cdist(blk.reshape(-1, 33), c).reshape(nb, B, P)           # |q - c|
                lb = (d.min(dim=1).values - r[None, :]).clamp(min=0)                  # per (block, leaf)
                u = nn_s[s * B:(s + nb) * B].reshape(nb, B).max(dim=1).values       # block's largest true NN distance
                work += ((lb <= u[:, None]).float() * size[None, :]).sum().item()
                work_slack += ((lb <= 1.5 * u[:, None] + 1.0).float() * size[None, :]).sum().item()
            tot = nblk * float(fb.shape[0])
            print(f"P={P} B={B}: unpruned work fraction {work / tot:.4f} (with slack U*1.5+1: {work_slack / tot:.4f})  "
                  f"leaf r median {r.median().item():.1f} max {r.max().item():.1f}  [{time.time() - t:.1f}s]", flush=True)


if __name__ == "__main__":
    main()
