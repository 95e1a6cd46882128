"""bench.py's own N-rank launcher (`python bench.py --gpus N` with no torch.distributed.run around it),
rehearsed on CPU: gloo, two ranks, the oracle standing in for the HIP engine (tests/cpu_evaluator.py).
What is checked is the plumbing a GPU node will run — N real processes, rank-sharded points, ONE
all-reduce of [grad | sums] per step, max-over-ranks timing, one JSON line from rank 0 with n_gpus = N —
and that a 2-rank run computes the same loss as one process over both shards."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--points", "48", "--no-cpu-baseline"]


def run_bench(extra, env_extra=None, timeout=600):
    env = dict(os.environ, PINN_BENCH_TEST_EVALUATOR="tests.cpu_evaluator:make", PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS + extra, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def json_line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout, out.stderr[-2000:])
    return json.loads(lines[0])


def test_launcher_starts_two_ranks_and_reports_them():
    out = run_bench(["--gpus", "2"])
    assert out.returncode == 0, out.stderr[-3000:]
    r = json_line(out)
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["global_points"] == 96 and r["config"]["points_per_gpu"] == 48
    assert r["config"]["allreduce_bytes"] == (29636 + 3) * 4
    assert r["value"] > 0 and r["scaling"] == "weak" and r["data"].startswith("INVALID")
    assert "rank 0/2" in out.stderr and "rank 1/2" in out.stderr          # two processes really ran
    assert "launcher: started 2 ranks" in out.stderr


def test_rank_count_mismatch_is_refused():
    """WORLD_SIZE from a launcher that disagrees with --gpus: never report a job under another name."""
    out = run_bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_launcher_refuses_more_gpus_than_visible():
    """Without the test evaluator the launcher counts devices first: this container has none."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PINN_BENCH_TEST_EVALUATOR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
