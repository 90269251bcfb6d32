"""Pair sharding across ranks + the single collective of the path (one all-gather of 96-byte per-pair records).

The reference processes scan pairs in a plain sequential loop (src/main.cpp:384-407): pairs are independent units,
so they shard embarrassingly -- pair p goes to rank p mod world -- and the only exchange is the final all-gather of
the per-pair result records.  torch.distributed is plumbing here: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo"
in the CPU tests.
"""
import numpy as np

RECORD_FLOATS = 24   # one 96-byte record = 24 four-byte words: 16 x f32 transform (column major) | i32 converged, i32 iterations,
                     # i32 n_inliers | f32 time_cs, f32 time_te | i32 pair id | 2 spare.  The integer words are carried as int32 BIT
                     # PATTERNS inside the float32 buffer (an all-gather copies bytes), so counts above 2^24 (max_iterations defaults
                     # to INT_MAX, include/config.h) survive exactly.
_I_CONV, _I_ITER, _I_INL, _F_TCS, _F_TTE, _I_PAIR = 16, 17, 18, 19, 20, 21


def shard_pairs(n_pairs, world, rank):
    """indices of the pairs owned by `rank` (round-robin, like the static shard of SURVEY 8e)."""
    return list(range(rank, n_pairs, world))


def pack_record(pair_id, T_colmajor16, converged, iterations, n_inliers, time_cs, time_te):
    rec = np.zeros(RECORD_FLOATS, np.float32)
    rec[:16] = np.asarray(T_colmajor16, np.float32).reshape(16)
    words = rec.view(np.int32)
    words[_I_CONV], words[_I_ITER], words[_I_INL], words[_I_PAIR] = int(converged), int(iterations), int(n_inliers), int(pair_id)
    rec[_F_TCS], rec[_F_TTE] = time_cs, time_te
    return rec


def record_pair_ids(records):
    """int32 pair ids of a [k, RECORD_FLOATS] float32 record array (-1 marks a padding record)."""
    return np.ascontiguousarray(records, np.float32).view(np.int32)[:, _I_PAIR]


def unpack_record(rec):
    rec = np.ascontiguousarray(rec, np.float32)
    words = rec.view(np.int32)
    return dict(T=rec[:16].reshape(4, 4).T.copy(), converged=int(words[_I_CONV]), iterations=int(words[_I_ITER]),
                n_inliers=int(words[_I_INL]), time_cs=float(rec[_F_TCS]), time_te=float(rec[_F_TTE]), pair_id=int(words[_I_PAIR]))


def gather_records(local_records, world, device=None, force=False):
    """local_records: [k, RECORD_FLOATS] tensor of 4-byte words (int32 views of the records; k equal on all ranks, pad with pair_id = -1).
    Returns the [world * k, RECORD_FLOATS] tensor of all ranks' records on every rank (one all_gather).  world == 1 needs no exchange;
    force=True still issues the collective on the (one-rank) process group -- bench.py --force-collective, which runs the RCCL code
    of the N > 1 path on a one-GPU box."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
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
    local.view(np.int32)[:, _I_PAIR] = -1
    for s, p in enumerate(mine):
        local[s] = align_fn(p)
    t = torch.from_numpy(local.view(np.int32))       # travels as int32 words: no float canonicalisation anywhere on the way
    if device is not None:
        t = t.to(device)
    allr = np.ascontiguousarray(gather_records(t, world).cpu().numpy()).view(np.float32)
    ids = record_pair_ids(allr)
    allr = allr[ids >= 0]
    return allr[np.argsort(ids[ids >= 0], kind="stable")]
