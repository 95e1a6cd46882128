"""RCCL + HIP evaluator on the ONE GPU a test box has: the data-parallel path — all-reduce of
[grad | loss sums], parameter broadcast, row gather — run for real on device tensors through torch.distributed's
"nccl" backend (RCCL on ROCm) in a world of one.  Each case is a child process with a fresh interpreter (a process
group must not leak into the pytest process; nothing here replaces a running program)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def _bench(extra_env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--points", "65536"], env=_env(**extra_env), cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_with_a_world_of_one_goes_through_rccl():
    plain = _bench({})
    forced = _bench({"PINN_BENCH_FORCE_DIST": "1"})
    P = plain["config"]["params"]
    assert plain["config"]["allreduce_bytes"] == 0 and plain["config"]["allreduce_ms"] == 0.0
    assert forced["config"]["allreduce_bytes"] == (P + 3) * 4          # [grad | three term sums], ONE collective per step
    assert forced["config"]["allreduce_ms"] > 0.0
    assert forced["n_gpus"] == 1 and forced["config"]["parallelism"] == "dp1"
    # summing over one rank changes nothing: the same trajectory (the tile kernel accumulates in lock order: 1e-6)
    a, b = plain["config"]["final_loss"], forced["config"]["final_loss"]
    assert abs(a - b) <= 2e-6 * abs(a), (a, b)


def test_launcher_counts_gpus_without_loading_hip():
    code = textwrap.dedent("""
        import sys, json
        sys.path.insert(0, %r)
        import bench
        n = bench.visible_gpu_count()
        assert "torch" not in sys.modules, "the launcher's count must not import torch"
        import torch
        print(json.dumps({"sysfs": n, "torch": torch.cuda.device_count()}))
    """ % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["sysfs"] == r["torch"] >= 1, r


def test_trainer_with_an_initialised_nccl_group(tmp_path):
    """trainer.PINN under an initialised world-1 "nccl" group with the collectives forced on: shard / broadcast /
    all-reduce / gather_rows / dump_predictions all run on device tensors, and the run equals the plain one."""
    code = textwrap.dedent("""
        import json, os, sys
        import numpy as np, torch
        import torch.distributed as dist
        sys.path.insert(0, %r)
        from pinn_depthestimation_amd.parallel import Reducer
        from pinn_depthestimation_amd.trainer import PINN
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        cfg = {"layers": {"input_features": 3, "hidden_layers": 8, "hidden_width": 64, "output_features": 4},
               "adam_optimizer": {"max_it": 20, "learning_rate": 1e-3, "scheduler_step_size": 10, "scheduler_gamma": 0.8},
               "lbfgs_optimizer": {"max_it": 0, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                                   "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
               "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
               "data_fidelity": {"inputs": ["t", "x", "y"], "outputs": []},
               "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "txy"}, "outputs": ["h", "z", "u", "v"]}}
        X = (torch.rand(5000, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1).numpy()
        out = {}
        for tag, red in (("plain", Reducer()), ("rccl", Reducer(force=True))):
            torch.manual_seed(1234)
            tr = PINN(None, None, X, cfg, reducer=red, log_every=1, checkpoint_every=0)
            assert tr.reducer.active == (tag == "rccl")
            tr.train()
            path = os.path.join(%r, tag + ".mat")
            tr.dump_predictions(path)
            from scipy.io import loadmat
            m = loadmat(path)
            out[tag] = {"losses": [h[3] for h in tr.history], "folded": tr._folded_iters,
                        "pred_h": float(np.abs(m["pred_h"]).sum()), "rows": int(m["pred_h"].shape[0])}
        g = Reducer(force=True).gather_rows(torch.arange(12, device="cuda", dtype=torch.float32).reshape(6, 2), 6)
        out["gather_ok"] = bool(torch.equal(g.cpu(), torch.arange(12, dtype=torch.float32).reshape(6, 2)))
        dist.destroy_process_group()
        print(json.dumps(out))
    """ % (ROOT, str(tmp_path)))
    res = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    r = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert r["gather_ok"]
    assert r["plain"]["folded"] == 20 and r["rccl"]["folded"] == 0       # the all-reduce sits between gradient and update
    assert r["plain"]["rows"] == r["rccl"]["rows"] == 5000
    a, b = r["plain"]["losses"], r["rccl"]["losses"]
    assert len(a) == len(b) == 20
    assert max(abs(x - y) / abs(x) for x, y in zip(a, b)) < 2e-5
    assert abs(r["plain"]["pred_h"] - r["rccl"]["pred_h"]) <= 1e-4 * abs(r["plain"]["pred_h"])


_TWO_RANK = """
import json, os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from pinn_depthestimation_amd.parallel import Reducer
from pinn_depthestimation_amd.trainer import PINN
torch.cuda.set_device(0)
world = int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo")          # both ranks drive cuda:0; gloo stages the device buffers through the host
cfg = {"layers": {"input_features": 3, "hidden_layers": 8, "hidden_width": 64, "output_features": 4},
       "adam_optimizer": {"max_it": 12, "learning_rate": 1e-3, "scheduler_step_size": 10, "scheduler_gamma": 0.8},
       "lbfgs_optimizer": {"max_it": 6, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                           "tolerance_grad": 1e-9, "tolerance_change": 1e-12, "line_search_fn": "strong_wolfe"},
       "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
       "data_fidelity": {"inputs": ["t", "x", "y"], "outputs": []},
       "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "txy"}, "outputs": ["h", "z", "u", "v"]}}
X = (torch.rand(5001, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1).numpy()     # odd: ragged shards
torch.manual_seed(1234 + int(os.environ["RANK"]))        # rank 1 starts from DIFFERENT weights: the broadcast from rank 0 must fix that
tr = PINN(None, None, X, cfg, reducer=Reducer(), log_every=1, checkpoint_every=0)
assert tr.reducer.active == (world > 1) and tr.reducer.world == world
tr.train()
theta = tr.theta.detach().double().cpu().numpy()
out = {"losses": [h[3] for h in tr.history], "theta_sum": float(theta.sum()), "theta_abs": float(np.abs(theta).sum()),
       "folded": tr._folded_iters}
if world > 1:
    both = [None, None]
    dist.all_gather_object(both, (out["theta_sum"], out["theta_abs"]))
    out["ranks_agree"] = both[0] == both[1]
    dist.destroy_process_group()
print(json.dumps(out))
"""


def test_two_ranks_on_one_gpu_equal_the_single_process_run():
    """World size 2 with the HIP evaluator on device tensors: two processes share the box's one GPU (RCCL refuses two
    ranks on one device, so the collectives go through gloo — the Reducer is backend-agnostic), each runs the fused
    kernel on its ragged shard, [grad | loss sums] is all-reduced per evaluation.  Adam + L-BFGS losses equal the
    single-process run on the whole set, and both ranks end on bit-identical parameters although they were
    initialised differently."""
    code = _TWO_RANK % {"root": ROOT}
    port = str(_port())
    procs = [subprocess.Popen([sys.executable, "-c", code], env=_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_PORT=port),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    one = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert outs[0]["ranks_agree"] and outs[1]["ranks_agree"]
    assert outs[0]["theta_sum"] == outs[1]["theta_sum"] and outs[0]["losses"] == outs[1]["losses"]
    assert outs[0]["folded"] == 0 and ref["folded"] == 12
    a, b = ref["losses"], outs[0]["losses"]
    assert len(a) == len(b) >= 13                       # 12 Adam iterations + the L-BFGS evaluations
    # Adam stage: the shard sums add up to the whole-set sums (fp32 association only)
    assert max(abs(x - y) / abs(x) for x, y in zip(a[:12], b[:12])) < 5e-5, (a[:12], b[:12])
    # L-BFGS amplifies last-bit differences in its line search: same descent, not the same digits
    assert abs(a[-1] - b[-1]) < 0.05 * abs(a[-1]), (a[-1], b[-1])
