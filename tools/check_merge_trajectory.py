import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pinn_depthestimation_amd.trainer import PINN
def cfg_of(hidden, width, n):
    return {"layers": {"input_features": 2, "hidden_layers": hidden, "hidden_width": width, "output_features": 6},
           "adam_optimizer": {"max_it": n, "learning_rate": 1e-4, "scheduler_step_size": 1000, "scheduler_gamma": 0.8},
           "lbfgs_optimizer": {"max_it": 0, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                               "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
           "loss": {"weight_fid_loss": 1, "weight_res_loss": 1, **{f"weight_{k}_loss": 1.0 for k in ("h", "U", "V", "eta_mean", "Hrms", "k")}},
           "data_fidelity": {"inputs": ["x", "y"], "outputs": ["h", "U", "V", "eta_mean", "Hrms", "k"]},
           "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "xy"},
                             "outputs": ["h", "U", "V", "eta_mean", "Hrms", "k"]}}
g = torch.Generator().manual_seed(7)
Xr = (torch.rand(243, 2, generator=g) * 2 - 1).numpy(); Xf = (torch.rand(12, 2, generator=g) * 2 - 1).numpy()
Tf = (torch.rand(12, 6, generator=g) * 0.2 + 0.7).numpy()
for hidden, width in ((10, 10), (8, 64)):
    runs = {}
    for merge in (False, True):
        torch.manual_seed(1234)
        tr = PINN(Xf, Tf, Xr, cfg_of(hidden, width, 300), log_every=1, checkpoint_every=0)
        tr.evaluator.merge_sets = merge
        tr.loss_func()
        g0 = tr.grad.clone()
        tr.iter = 0; tr.history = []
        tr.train()
        runs[merge] = (np.array([h[1:] for h in tr.history]), g0)
    a, b = runs[False], runs[True]
    rel = np.abs(a[0] - b[0]) / np.abs(a[0])
    print(f"{hidden}x{width}: grad0 rel-L2 diff {float((a[1]-b[1]).norm()/a[1].norm()):.2e}; loss rel diff step1 {rel[0]}, step10 {rel[9]}, step100 {rel[99]}, step300 {rel[-1]}; losses@300 {a[0][-1]} {b[0][-1]}")
