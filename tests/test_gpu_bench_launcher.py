"""`python bench.py --gpus N` outside torch.distributed.run launches its own N ranks (children spawned before anything
touches the GPU; the parent relays rank 0's line and fails if a rank fails) -- VERDICT r2 #3.  Two ranks share the one
GPU of the box here (`--same-device`) and exchange their spike windows through host memory (`--exchange host`): the
product's sharded sim() with the gloo control plane, end to end, as the driver's 2/4/8-GPU runs start it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "host", "--same-device", "--workload", "c4",
           "--tiles-per-gpu", "32", "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["neurons"] == 2 * 32 * 256 and d["config"]["exchange"].startswith("host all-gather")
    assert d["totals_in_timed_region"]["neurons_updated"] == 8 * 2 * 32 * 256  # the totals of the WHOLE chip


def test_bench_fails_when_a_rank_fails():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "host", "--same-device", "--workload", "c2",
           "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]  # c2 is a single-GPU configuration: every rank exits non-zero
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
