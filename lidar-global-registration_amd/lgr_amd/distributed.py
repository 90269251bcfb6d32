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


def assign_pairs(n_pairs, world, policy="round_robin", costs=None):
    """owner rank of every pair -- identical on every rank (pure function of its arguments, no communication).
      round_robin  pair p -> rank p mod world: the static shard of SURVEY 8(e);
      lpt          longest processing time first, SURVEY 8(e)'s "greedy by point count": pairs in order of decreasing predicted cost
                   (ties: lower pair id) each go to the rank with the least load so far (ties: lower rank).  costs[p] = predicted cost of
                   pair p, e.g. pair_cost(M_src, M_tgt).  With the size mix of data/tests.yaml (the brute-force matcher makes a 1M-point pair
                   cost ~10 x a 100 k one) round-robin's makespan is whatever rank draws the most large pairs; LPT's is within 4/3 of optimal."""
    if policy == "round_robin":
        return [p % world for p in range(n_pairs)]
    if policy != "lpt":
        raise ValueError("policy must be 'round_robin' or 'lpt'")
    if costs is None or len(costs) != n_pairs:
        raise ValueError("lpt needs one predicted cost per pair")
    load = [0.0] * world
    owner = [0] * n_pairs
    for p in sorted(range(n_pairs), key=lambda q: (-float(costs[q]), q)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[p] = r
        load[r] += float(costs[p])
    return owner


def shard_pairs(n_pairs, world, rank, policy="round_robin", costs=None):
    """indices of the pairs owned by `rank`, ascending (assign_pairs)."""
    return [p for p, r in enumerate(assign_pairs(n_pairs, world, policy, costs)) if r == rank]


def rank_order(n_pairs, world, rank, policy="round_robin", costs=None):
    """the pairs of `rank` in the order run_pairs aligns them: with predicted costs largest first (the context's workspace -- grown on demand,
    never shrunk -- then reaches its final size with the first pair instead of being re-allocated at every new record size: up to 80 ms on such
    a pair in profiles/r5_job_tests156.json), ascending pair id otherwise.  Records are keyed by pair id: the order is invisible to the caller."""
    mine = shard_pairs(n_pairs, world, rank, policy, costs)
    if costs is not None:
        mine.sort(key=lambda q: (-float(costs[q]), q))
    return mine


def pair_cost(m_src, m_tgt):
    """predicted seconds of aligning a pair on one MI355X: the brute-force matcher's M_src * M_tgt term (include/matching.h:594-634; 16 of a
    1M-point pair's 24.5 ms) + a per-point term for the feature stages + a fixed part (RANSAC, launches); fitted to profiles/r5_job_tests156.json.
    Only the ORDER of the costs matters to assign_pairs."""
    return 1.6e-14 * float(m_src) * float(m_tgt) + 3.5e-9 * (float(m_src) + float(m_tgt)) + 1.5e-3


def makespan(times, world, policy="round_robin", costs=None):
    """max over ranks of the summed `times` of the pairs a policy assigns to it (times: measured or predicted seconds per pair)"""
    load = [0.0] * world
    for p, r in enumerate(assign_pairs(len(times), world, policy, costs)):
        load[r] += float(times[p])
    return max(load) if load else 0.0


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


def run_pairs(n_pairs, world, rank, align_fn, device=None, policy="round_robin", costs=None):
    """Process this rank's shard with align_fn(pair_id) -> record (numpy [RECORD_FLOATS]); all-gather; return the
    records of all pairs ordered by pair id (numpy [n_pairs, RECORD_FLOATS]).  policy / costs: assign_pairs."""
    import torch
    owner = assign_pairs(n_pairs, world, policy, costs)
    mine = rank_order(n_pairs, world, rank, policy, costs)
    k = max([owner.count(r) for r in range(world)] + [0])      # shards are padded to the largest one (equal counts per rank for the all-gather)
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
