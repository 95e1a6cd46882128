#!/usr/bin/env python3
"""newmethod_latency.py — Adam iteration time of the train_newmethod.py problem shape: 2->100x20->3,
continuity_only + F.mse_loss on U, V over the SAME 12 514 points (data_at50k.mat), one combined pass."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pinn_depthestimation_amd.trainer import PINN

cfg = {"layers": {"input_features": 2, "hidden_layers": 100, "hidden_width": 20, "output_features": 3,
                  "dropout_rate": 0.0, "init_type": "xavier"},
       "adam_optimizer": {"max_it": 10, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
       "lbfgs_optimizer": {"max_it": 0},
       "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
       "data": {"inputs": {"x": {"requires_grad": ["true"]}, "y": {"requires_grad": ["true"]}},
                "trues": ["U", "V"], "unknowns": ["h"]}}
g = torch.Generator().manual_seed(5)
for n in (12514, 200000):
    X = (torch.rand(n, 2, generator=g) * 2 - 1).numpy()
    T = (torch.rand(n, 2, generator=g) * 0.4 - 0.2).numpy()
    torch.manual_seed(1234)
    tr = PINN(X, T, X, cfg, log_every=1, checkpoint_every=0)
    for _ in range(10):
        tr.adam_step()
    torch.cuda.synchronize()
    steps = 200
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.adam_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"newmethod 100x20 N={n}: {dt / steps * 1e6:9.1f} us/step ({n * steps / dt:.3e} points/s) loss {tr.last[2].item():.5e}")
