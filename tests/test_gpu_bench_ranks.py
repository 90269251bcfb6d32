"""bench.py --gpus N on real hardware (-m gpu): the parent starts the ranks itself (no launcher in the environment), every rank
registers its own scan pair on the card and the 96-byte records are all-gathered.  A one-GPU box cannot give every rank its own
device, so the collective runs over gloo and both ranks share cuda:0 (`--backend gloo`, bench.py's rehearsal mode); the rank
formation, sharding, timing protocol (barrier + max over ranks) and the JSON line are the ones the 8-GPU run uses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_one_card():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--points", "200000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE line
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["pairs_per_step"] == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]   # whole-job rate: both ranks' pairs
    assert d["result"]["converged"] == 1


def test_bench_force_collective_runs_rccl_on_one_rank():
    """the RCCL calls of the N > 1 path -- init_process_group("nccl", device_id), the all-gather of the DEVICE
    record tensor, barrier, all_reduce(MAX) of the elapsed time -- executed on the one-GPU box through a one-rank group, in a fresh
    child process (as the driver starts bench.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "2", "--warmup", "1",
                        "--points", "200000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 1 and d["collective"]["process_group"] is True and d["collective"]["backend"] == "nccl"
    assert d["collective"]["all_gathers_executed"] == 3 and d["collective"]["records_per_gather"] == 1      # warmup + steps
    assert d["result"]["converged"] == 1 and d["value"] > 0
    assert 1 <= d["host"]["threads_per_rank"] <= 3 and d["host"]["usable_cores"] >= 1


def test_bench_single_context_mode():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-context", "--steps", "2", "--warmup", "1",
                        "--points", "200000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][0]
    assert d["host"]["threads_per_rank"] == 1 and d["host"]["helper_contexts"] == 0 and d["result"]["converged"] == 1
