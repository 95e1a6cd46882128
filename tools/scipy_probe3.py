import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from threadpoolctl import threadpool_limits, threadpool_info
from pinn_depthestimation_amd.trainer import PINN
from pinn_depthestimation_amd.lbfgsb import LBFGSBOptimizer
N = 1 << 20
cfg = {"layers": {"input_features": 3, "hidden_layers": 8, "hidden_width": 64, "output_features": 4},
       "adam_optimizer": {"max_it": 50, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
       "lbfgs_optimizer": {"max_it": 15, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                           "tolerance_grad": 0.0, "tolerance_change": 0.0, "line_search_fn": "strong_wolfe"},
       "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
       "data_fidelity": {"inputs": ["t", "x", "y"], "outputs": []},
       "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "txy"}, "outputs": ["h", "z", "u", "v"]}}
X = (torch.rand(N, 3, generator=torch.Generator().manual_seed(1234)) * 2 - 1).numpy()
print([(d["user_api"], d["internal_api"], d["num_threads"]) for d in threadpool_info()], "torch threads", torch.get_num_threads())
def T(): torch.cuda.synchronize(); return time.perf_counter()
for mode in ("default", "blas/openmp limited to 1", "torch.set_num_threads(1)"):
    torch.manual_seed(1234)
    tr = PINN(None, None, X, cfg, log_every=1000, checkpoint_every=0)
    tr.train_adam(50); torch.cuda.synchronize()
    opt = LBFGSBOptimizer(tr, {"maxiter": 15, "maxfun": 150, "ftol": 0.0, "gtol": 0.0})
    orig = opt.function_for_scipy; log = []
    def timed(x, orig=orig, log=log):
        t0 = T(); r = orig(x); log.append(T() - t0); return r
    opt.function_for_scipy = timed
    t0 = T()
    if mode.startswith("blas"):
        with threadpool_limits(limits=1):
            opt.minimize()
    else:
        if mode.startswith("torch"): torch.set_num_threads(1)
        opt.minimize()
    print(f"{mode}: minimize {1e3*(T()-t0):.1f} ms; evaluations (ms): {[round(1e3*d,1) for d in log]}")
