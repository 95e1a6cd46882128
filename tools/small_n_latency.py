#!/usr/bin/env python3
"""small_n_latency.py — per-step wall time of the Adam loop at the reference's own problem sizes
(N_res = 243 as configured, 10 000 = BASELINE configs[0]), where launches and host work dominate:
trainer.PINN.adam_step with per-iteration logging (the reference's behaviour: one sync per step)
and with logging every 1000 steps (fully asynchronous)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pinn_depthestimation_amd.trainer import PINN


def ns_config(adam_it):
    return {
        "layers": {"input_features": 3, "hidden_layers": 8, "hidden_width": 64, "output_features": 4},
        "adam_optimizer": {"max_it": adam_it, "learning_rate": 1e-4, "scheduler_step_size": 1000, "scheduler_gamma": 0.8},
        "lbfgs_optimizer": {"max_it": 0, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                            "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
        "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
        "data_fidelity": {"inputs": ["t", "x", "y"], "outputs": []},
        "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "txy"}, "outputs": ["h", "z", "u", "v"]},
    }


def main():
    steps = int(os.environ.get("STEPS", "2000"))
    for n in (243, 10000, 100000):
        X = (torch.rand(n, 3, generator=torch.Generator().manual_seed(1234)) * 2 - 1).numpy()
        for log_every in (1, 1000):
            torch.manual_seed(1234)
            tr = PINN(None, None, X, ns_config(steps), log_every=log_every, checkpoint_every=0)
            run = getattr(tr, "train_adam", None)
            for _ in range(20):
                tr.adam_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if run is not None and os.environ.get("USE_RUN", "1") == "1":
                run(steps)
            else:
                for _ in range(steps):
                    tr.adam_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"N={n:7d} log_every={log_every:5d}: {dt / steps * 1e6:8.1f} us/step  ({n * steps / dt:.3e} points/s)  loss {tr.last[2].item():.6e}")


if __name__ == "__main__":
    main()


def cmb_like():
    """config_CMB.json's shape of problem: 2->10x10->6 (as written) and 2->8x64->6, physics_equation on 243
    collocation points + 12 fidelity points (6 weighted outputs), one launch vs two per iteration."""
    import json
    for hidden, width in ((10, 10), (8, 64)):
        cfg = {"layers": {"input_features": 2, "hidden_layers": hidden, "hidden_width": width, "output_features": 6},
               "adam_optimizer": {"max_it": 10, "learning_rate": 1e-4, "scheduler_step_size": 1000, "scheduler_gamma": 0.8},
               "lbfgs_optimizer": {"max_it": 0, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                                   "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
               "loss": {"weight_fid_loss": 1, "weight_res_loss": 1, **{f"weight_{k}_loss": 1.0 for k in ("h", "U", "V", "eta_mean", "Hrms", "k")}},
               "data_fidelity": {"inputs": ["x", "y"], "outputs": ["h", "U", "V", "eta_mean", "Hrms", "k"]},
               "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "xy"},
                                 "outputs": ["h", "U", "V", "eta_mean", "Hrms", "k"]}}
        g = torch.Generator().manual_seed(7)
        Xr = (torch.rand(243, 2, generator=g) * 2 - 1).numpy()
        Xf = (torch.rand(12, 2, generator=g) * 2 - 1).numpy()
        Tf = (torch.rand(12, 6, generator=g) * 0.2 + 0.7).numpy()
        for merge in (False, True):
            torch.manual_seed(1234)
            tr = PINN(Xf, Tf, Xr, cfg, log_every=1, checkpoint_every=0)
            tr.evaluator.merge_sets = merge
            for _ in range(20):
                tr.adam_step()
            torch.cuda.synchronize()
            steps = int(os.environ.get("STEPS", "2000"))
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.adam_step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"CMB-like {hidden}x{width} N_res=243 N_fid=12 merge={merge}: {dt / steps * 1e6:8.1f} us/step  loss {tr.last[2].item():.6e}")


if __name__ == "__main__" and os.environ.get("CMB", "1") == "1":
    cmb_like()
