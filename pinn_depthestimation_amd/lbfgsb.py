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

    def _staging(self, P: int, device):
        """Pinned host buffers, allocated once: x (fp32) on its way in, [gradient | loss] on its way out.  With pageable
        memory every evaluation paid three blocking copies (x in, float(loss), gradient out: 25 ms around a 6.5 ms
        closure at N = 2^20, tools/lbfgs_stage_profile.py); pinned and asynchronous it is ONE synchronisation."""
        st = getattr(self, "_stage", None)
        if st is None or st[0].numel() != P:
            pin = device.type == "cuda"
            st = (torch.empty(P, dtype=torch.float32, pin_memory=pin), torch.empty(P + 1, dtype=torch.float32, pin_memory=pin))
            self._stage = st
        return st

    def function_for_scipy(self, x: np.ndarray):
        """flat float64 vector -> (loss, flat float64 gradient); one closure evaluation."""
        tr = self.trainer
        theta = tr.dnn.flat_params()
        P = theta.numel()
        x32, out = self._staging(P, theta.device)
        x32.copy_(torch.from_numpy(np.asarray(x, dtype=np.float64)))          # fp64 -> fp32 on the host, into pinned memory
        theta.copy_(x32, non_blocking=True)
        loss = tr.loss_func()
        out[:P].copy_(tr.grad.detach(), non_blocking=True)
        out[P:].copy_(loss.detach().reshape(1), non_blocking=True)
        if theta.is_cuda:
            torch.cuda.current_stream(theta.device).synchronize()
        f = float(out[P])
        self.losses.append(f)
        return f, out[:P].double().numpy().copy()

    def minimize(self):
        from scipy.optimize import minimize
        x0 = self.trainer.dnn.flat_params().detach().to("cpu", torch.float64).numpy()
        # SciPy's L-BFGS-B does its level-1 BLAS on P-vectors through OpenBLAS, whose pool sizes itself by the HOST's
        # core count (64 threads on the MI355X boxes, 16 cores granted): its spinning workers stalled every third
        # evaluation for ~80 ms (tools/scipy_probe3.py: 18 evaluations in 1034 ms, 135 ms with the pool limited to one
        # thread — 30 k-element dot products gain nothing from threads)
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1, user_api="blas")
        except Exception:          # pragma: no cover  (threadpoolctl absent: run as is)
            import contextlib
            limit = contextlib.nullcontext()
        with limit:
            res = minimize(self.function_for_scipy, x0, jac=True, method="L-BFGS-B", options=self.options)
        self.function_for_scipy(res.x)      # leave the best point in the network
        self.trainer.flush_log()
        return res
