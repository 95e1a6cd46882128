"""lbfgsb — SciPy L-BFGS-B fine-tune stage (BASELINE configs[4]).

The reference tree holds only stale bytecode of a TensorFlow-era `l_bfgs_b_optimizer`
(__pycache__/l_bfgs_b_optimizer.cpython-310.pyc; no source, nothing imports it — SURVEY.md
fact 0.4).  Its visible contract: LBFGSBOptimizer(model, inputs, outputs, loss_function,
options=None) wrapping scipy.optimize.minimize(fun, x0, jac=True, method='L-BFGS-B',
options={maxiter 50000, maxfun 50000, maxcor 50, maxls 50, ftol 1.0*np.finfo(float).eps})
over the flattened weights.  Here the same driver runs over the trainer's flat closure:
x (float64, host) -> theta (fp32, device) -> one fused loss+grad evaluation -> (loss, grad).
"""
from __future__ import annotations

import numpy as np
import torch

DEFAULT_OPTIONS = {"maxiter": 50000, "maxfun": 50000, "maxcor": 50, "maxls": 50,
                   "ftol": 1.0 * np.finfo(float).eps}


class LBFGSBOptimizer:
    def __init__(self, trainer, options=None):
        self.trainer = trainer
        self.options = dict(DEFAULT_OPTIONS)
        if options:
            self.options.update(options)
        self.losses = []

    def function_for_scipy(self, x: np.ndarray):
        """flat float64 vector -> (loss, flat float64 gradient); one closure evaluation."""
        tr = self.trainer
        theta = tr.dnn.flat_params()
        theta.copy_(torch.from_numpy(np.asarray(x, dtype=np.float64)).to(theta.device, torch.float32))
        loss = tr.loss_func()
        self.losses.append(float(loss))
        return float(loss), tr.grad.detach().to("cpu", torch.float64).numpy().copy()

    def minimize(self):
        from scipy.optimize import minimize
        x0 = self.trainer.dnn.flat_params().detach().to("cpu", torch.float64).numpy()
        res = minimize(self.function_for_scipy, x0, jac=True, method="L-BFGS-B", options=self.options)
        self.function_for_scipy(res.x)      # leave the best point in the network
        self.trainer.flush_log()
        return res
