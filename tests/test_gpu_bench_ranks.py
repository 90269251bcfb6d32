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
