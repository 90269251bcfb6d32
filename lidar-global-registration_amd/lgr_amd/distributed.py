"""Pair sharding across ranks + the single collective of the path (one all-gather of 96-byte per-pair records).

The reference processes scan pairs in a plain sequential loop (src/main.cpp:384-407): pairs are independent units,
so they shard embarrassingly -- pair p goes to rank p mod world -- and the only exchange is the final all-gather of
the per-pair result records.  torch.distributed is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo"
in the CPU tests.
"""
import numpy as np

RECORD_FLOATS = 24   # 16 transform + converged, iterations, n_inliers, time_cs, time_te, pair id, 2 spare = 96 bytes


def shard_pairs(n_pairs, world, rank):
    """indices of the pairs owned by `rank` (round-robin, like the static shard of SURVEY 8e)."""
    return list(range(rank, n_pairs, world))


def pack_record(pair_id, T_colmajor16, converged, iterations, n_inliers, time_cs, time_te):
    rec = np.zeros(RECORD_FLOATS, np.float32)
    rec[:16] = np.asarray(T_colmajor16, np.float32).reshape(16)
    rec[16:22] = [converged, iterations, n_inliers, time_cs, time_te, pair_id]
    return rec


def unpack_record(rec):
    rec = np.asarray(rec, np.float32)
    return dict(T=rec[:16].reshape(4, 4).T.copy(), converged=int(rec[16]), iterations=int(rec[17]), n_inliers=int(rec[18]),
                time_cs=float(rec[19]), time_te=float(rec[20]), pair_id=int(rec[21]))


def gather_records(local_records, world, device=None):
    """local_records: [k, RECORD_FLOATS] float32 tensor (k equal on all ranks, pad with pair_id = -1).
    Returns the [world * k, RECORD_FLOATS] tensor of all ranks' records on every rank (one all_gather)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_records
    out = [torch.empty_like(local_records) for _ in range(world)]
    dist.all_gather(out, local_records)
    return torch.cat(out, 0)


def run_pairs(n_pairs, world, rank, align_fn, device=None):
    """Process this rank's shard with align_fn(pair_id) -> record (numpy [RECORD_FLOATS]); all-gather; return the
    records of all pairs ordered by pair id (numpy [n_pairs, RECORD_FLOATS])."""
    import torch
    mine = shard_pairs(n_pairs, world, rank)
    k = (n_pairs + world - 1) // world
    local = np.zeros((k, RECORD_FLOATS), np.float32)
    local[:, 21] = -1
    for s, p in enumerate(mine):
        local[s] = align_fn(p)
    t = torch.from_numpy(local)
    if device is not None:
        t = t.to(device)
    allr = gather_records(t, world).cpu().numpy()
    allr = allr[allr[:, 21] >= 0]
    return allr[np.argsort(allr[:, 21], kind="stable")]
