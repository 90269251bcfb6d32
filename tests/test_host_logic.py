"""Host-side logic that needs no GPU: synthetic generator, pair sharding, record packing, and the N > 1 path
(world_size 2 over gloo): shard -> per-pair record -> ONE all_gather -> every rank holds every pair's record."""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lidar-global-registration_amd"))
from lgr_amd import distributed, synthetic  # noqa: E402


def test_synthetic_pair_is_deterministic_and_consistent():
    a = synthetic.make_pair(5000, seed=9)
    b = synthetic.make_pair(5000, seed=9)
    np.testing.assert_array_equal(a["src"], b["src"])
    np.testing.assert_array_equal(a["tgt"], b["tgt"])
    assert a["src"].shape == (5000, 12) and a["src"].dtype == np.float32
    assert (a["src"][:, 3] == 1).all() and (a["src"][:, 8] == 1).all() and (a["src"][:, 4:8] == 0).all()
    T = a["T_gt"]
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12) and np.linalg.det(T[:3, :3]) > 0
    # the overlap really overlaps: tgt mapped back to the scene frame shares the x range [1/3, 2/3] with src
    back = (a["tgt"][:, :3].astype(np.float64) - T[:3, 3]) @ T[:3, :3]
    lx = 24.0 * a["scale"]
    assert back[:, 0].min() < 0.4 * lx and a["src"][:, 0].max() > 0.6 * lx


def test_correspondence_problem_inlier_fraction():
    pr = synthetic.make_correspondence_problem(n_pts=4000, c=1000, inlier_frac=0.4, sigma=0.001, thr=0.05, seed=2)
    s = pr["src"][pr["corr"]["index_query"], :3].astype(np.float64)
    t = pr["tgt"][pr["corr"]["index_match"], :3].astype(np.float64)
    d = np.linalg.norm(s @ pr["T_gt"][:3, :3].T + pr["T_gt"][:3, 3] - t, axis=1)
    assert abs((d < 0.05).mean() - 0.4) < 0.03


@pytest.mark.parametrize("n,world", [(156, 8), (7, 2), (3, 4), (0, 2)])
def test_shard_pairs_partition(n, world):
    shards = [distributed.shard_pairs(n, world, r) for r in range(world)]
    flat = sorted(x for s in shards for x in s)
    assert flat == list(range(n))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def _job_costs(n=156):
    import importlib.util, os as _os
    spec = importlib.util.spec_from_file_location("bench_mod", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    sizes = b.job_sizes(n)
    return sizes, [distributed.pair_cost(m, m) for m in sizes]


@pytest.mark.parametrize("world", [2, 8])
def test_lpt_sharding_partitions_and_beats_round_robin_on_the_tests_yaml_size_mix(world):
    """"pair p -> GPU p mod G (or greedy by point count)".  On the 156-pair job's size mix (1e5 .. 1e6 points, cost ~ M_src M_tgt)
    the LPT shard's makespan is never above round-robin's, it is deterministic, and it partitions the pairs."""
    sizes, costs = _job_costs()
    assert min(sizes) >= 100000 and max(sizes) <= 1000000 and len(set(sizes)) > 20
    own = distributed.assign_pairs(len(costs), world, "lpt", costs)
    assert own == distributed.assign_pairs(len(costs), world, "lpt", list(costs))          # pure function: every rank computes the same owners
    shards = [distributed.shard_pairs(len(costs), world, r, "lpt", costs) for r in range(world)]
    assert sorted(x for s in shards for x in s) == list(range(len(costs))) and all(s == sorted(s) for s in shards)
    rr, lpt = distributed.makespan(costs, world, "round_robin"), distributed.makespan(costs, world, "lpt", costs)
    assert lpt <= rr and lpt <= 1.02 * sum(costs) / world + max(costs) * 0.34                # LPT bound: within 4/3 of the optimum
    # an adversarial order for round-robin: every world-th pair large
    adv = [distributed.pair_cost(1_000_000, 1_000_000) if p % world == 0 else distributed.pair_cost(100_000, 100_000) for p in range(64)]
    assert distributed.makespan(adv, world, "lpt", adv) < 0.6 * distributed.makespan(adv, world, "round_robin")
    with pytest.raises(ValueError):
        distributed.assign_pairs(4, 2, "lpt", None)
    with pytest.raises(ValueError):
        distributed.assign_pairs(4, 2, "random")


def test_record_roundtrip():
    T = synthetic.random_se3(np.random.default_rng(0)).astype(np.float32)
    rec = distributed.pack_record(17, T.T.reshape(16), 1, 123456, 789, 0.5, 0.25)
    assert rec.nbytes == 96
    u = distributed.unpack_record(rec)
    np.testing.assert_array_equal(u["T"], T)
    assert (u["converged"], u["iterations"], u["n_inliers"], u["pair_id"]) == (1, 123456, 789, 17)
    # integers above 2^24 (max_iterations defaults to INT_MAX) are carried as int32 bit patterns, not rounded through float32
    big = distributed.unpack_record(distributed.pack_record(2**24 + 1, T.T.reshape(16), 0, 2**31 - 1, 2**30 + 3, 1.5, 2.5))
    assert (big["iterations"], big["n_inliers"], big["pair_id"]) == (2**31 - 1, 2**30 + 3, 2**24 + 1)
    assert (big["time_cs"], big["time_te"]) == (1.5, 2.5)


def _fake_align(pair_id):
    T = synthetic.random_se3(np.random.default_rng(1000 + pair_id)).astype(np.float32)
    return distributed.pack_record(pair_id, T.T.reshape(16), pair_id % 2, 2**31 - 1 - 10 * pair_id, pair_id + 5, 0.1, 0.2)


def _worker(rank, world, port, n_pairs, q, policy="round_robin"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        costs = [1.0 + (7 * p) % 5 for p in range(n_pairs)] if policy == "lpt" else None     # uneven shards: 4 / 3 pairs, LPT's own grouping
        out = distributed.run_pairs(n_pairs, world, rank, _fake_align, policy=policy, costs=costs)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("policy", ["round_robin", "lpt"])
def test_two_ranks_gloo_allgather(policy):
    """world-size-2 gloo: shard (round-robin, and LPT with uneven shards) -> per-pair records -> ONE all_gather -> every rank holds every
    pair's record in pair order, whatever rank aligned it"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_pairs, world = 7, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q, policy)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([_fake_align(i) for i in range(n_pairs)])
    for r in range(world):
        np.testing.assert_array_equal(got[r], want)      # every rank holds every pair's record, ordered by pair id


def test_single_rank_run_pairs():
    out = distributed.run_pairs(5, 1, 0, _fake_align)
    np.testing.assert_array_equal(out, np.stack([_fake_align(i) for i in range(5)]))


def _run_bench(*argv, env_extra=None):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_2_forms_two_ranks():
    """`python bench.py --gpus 2` (no launcher) must start 2 ranks itself and report n_gpus = 2 (pairs shard,
    one all-gather of the 96-byte records).  --dry-run keeps the GPU out of it; the collective runs over gloo."""
    r, out = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1                              # rank 0 prints ONE line
    assert out[0]["n_gpus"] == 2 and out[0]["ranks_seen"] == [0, 1] and out[0]["records_ok"] is True


def test_bench_gpus_8_dry_run_shards_156_pairs():
    """The 8-GPU job shape without hardware (data/tests.yaml = 156 pairs, src/main.cpp:384-407): `bench.py --gpus 8 --dry-run --pairs 156`
    forms 8 gloo ranks, every rank runs its shard of distributed.run_pairs (uneven: 20 / 19 pairs), one all-gather of the padded shards, and rank 0
    holds all 156 records in pair order."""
    r, out = _run_bench("--gpus", "8", "--dry-run", "--pairs", "156", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1 and out[0]["n_gpus"] == 8 and out[0]["ranks_seen"] == list(range(8)) and out[0]["records_ok"] is True
    assert out[0]["pairs"] == 156 and out[0]["shard_sizes"] == [20, 20, 20, 20, 19, 19, 19, 19]


def test_bench_rejects_world_size_mismatch():
    r, out = _run_bench("--gpus", "1", "--dry-run", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not out


def test_bench_force_collective_one_rank_group():
    """--force-collective: a one-rank process group is formed without a launcher and the record all-gather goes through it (gloo here;
    the -m gpu twin in tests/test_gpu_bench_ranks.py runs the same code on RCCL)."""
    r, out = _run_bench("--gpus", "1", "--dry-run", "--force-collective", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1 and out[0]["n_gpus"] == 1 and out[0]["ranks_seen"] == [0] and out[0]["records_ok"] is True
    assert out[0]["collective"] == {"process_group": True, "backend": "gloo"}


def test_rank_order_largest_first_and_records_by_pair_id():
    """run_pairs aligns a rank's pairs in order of decreasing predicted cost (the workspace then grows once), records come back by pair id."""
    from lgr_amd import distributed
    sizes = [3, 9, 1, 7, 5, 8, 2]
    costs = [distributed.pair_cost(1000 * n, 1000 * n) for n in sizes]
    assert distributed.rank_order(len(sizes), 1, 0, "lpt", costs) == [1, 5, 3, 4, 0, 6, 2]
    assert distributed.rank_order(len(sizes), 1, 0) == list(range(len(sizes)))
    seen = []

    def fake(p):
        seen.append(p)
        return distributed.pack_record(p, np.eye(4, dtype=np.float32).T.reshape(16), 1, 10 * p, p, 0.0, 0.0)
    out = distributed.run_pairs(len(sizes), 1, 0, fake, policy="lpt", costs=costs)
    assert seen == [1, 5, 3, 4, 0, 6, 2]
    assert [distributed.unpack_record(r)["pair_id"] for r in out] == list(range(len(sizes)))
    for w in (2, 3):   # the ranks' orders partition the pairs
        allp = sorted(p for r in range(w) for p in distributed.rank_order(len(sizes), w, r, "lpt", costs))
        assert allp == list(range(len(sizes)))
