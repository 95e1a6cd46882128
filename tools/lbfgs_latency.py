#!/usr/bin/env python3
"""lbfgs_latency.py — wall time per torch.optim.LBFGS iteration (history 100, strong Wolfe: the
reference's settings, train.py:116-125) against the closure time at the reference's problem sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pinn_depthestimation_amd.trainer import PINN
from small_n_latency import ns_config


def main():
    for n, impl in ((243, "flat"), (243, "torch"), (243, "flat"), (10000, "torch"), (10000, "flat"), (1 << 20, "torch"), (1 << 20, "flat")):
        X = (torch.rand(n, 3, generator=torch.Generator().manual_seed(1234)) * 2 - 1).numpy()
        iters = 60 if n < 100000 else 15
        cfg = ns_config(50)
        cfg["lbfgs_optimizer"].update({"max_it": iters, "tolerance_grad": 0.0, "tolerance_change": 0.0, "history_size": 100})
        torch.manual_seed(1234)
        tr = PINN(None, None, X, cfg, log_every=1, checkpoint_every=0, lbfgs_impl=impl)
        for _ in range(50):
            tr.adam_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.closure()
        torch.cuda.synchronize()
        t_closure = (time.perf_counter() - t0) / 20
        evals0 = tr.iter
        t0 = time.perf_counter()
        tr.optimizer_LBFGS.step(tr.closure)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        evals = tr.iter - evals0
        n_it = tr.optimizer_LBFGS.state_dict()["state"][0]["n_iter"]
        print(f"N={n:8d} {impl:5s}: loss {tr.last[2].item():.6e} closure {t_closure * 1e6:8.1f} us | LBFGS {n_it} iterations, {evals} closure evals in {dt * 1e3:8.1f} ms "
              f"= {dt / max(n_it, 1) * 1e3:6.2f} ms/iteration ({dt / max(evals, 1) * 1e6:8.1f} us per closure eval incl. optimizer math)")


if __name__ == "__main__":
    main()
