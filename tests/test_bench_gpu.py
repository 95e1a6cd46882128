"""bench.py's extra workloads on a GPU (small sizes): the JSON contract of the narrow-network lines (roofline object, PMC
traffic source) and of the L-BFGS stage line (BASELINE configs[4]: both drivers, closure evaluations counted)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PINN_BENCH_FORCE_DIST"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["pe10x10", "co100x20"])
def test_narrow_network_lines(workload):
    r = _run(["--workload", workload, "--points", "16384", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["dtype"] == "f32" and r["value"] > 0
    ro = r["roofline"]
    assert ro["bound"] == "mfma" and ro["peak"] == 157.3 and 0 < ro["frac"] < 1 and ro["kernel_ms"] > 0
    assert ro["traffic"] and "profiles/r03" in ro["traffic_source"]      # bytes per point from the committed PMC passes
    assert f"[{workload}]" in r["config"]["workload"]


def test_lbfgs_stage_line():
    r = _run(["--workload", "lbfgs8x64", "--points", "16384", "--steps", "4"])
    assert r["unit"] == "residual-points/s" and r["value"] > 0 and "configs[4]" in r["config"]["workload"]
    d = r["config"]["drivers"]
    assert set(d) == {"torch.optim.LBFGS", "scipy L-BFGS-B"}
    for k, v in d.items():
        assert v["closure_evals"] >= 4 and v["seconds"] > 0 and 0 <= v["share_outside_closure"] < 1, (k, v)
        assert v["final_loss"] == v["final_loss"] and v["final_loss"] > 0          # finite, positive
