"""lbfgs — torch.optim.LBFGS semantics (the optimiser of train.py:116-125, one `.step(closure)`,
train.py:200) with the two-loop recursion done in six matrix-vector products.

torch.optim.LBFGS walks its history with two Python loops: at history_size = 100 that is ~400
tiny kernels (dot, axpy) per iteration — 2.3 ms of launches around a 0.15 ms closure at the
reference's problem sizes (measured, tools/lbfgs_latency.py).  The recursion is a pair of
triangular solves in disguise.  With S, Y the (k x P) matrices of steps and gradient
differences (oldest first), M = S Y^T and q0 = -g:

    first loop :  al_i = ro_i (s_i . q_i),  q_{i} = q0 - sum_{j>i} al_j y_j,  ro_i = 1 / M_ii
                  <=>  triu(M) al = S q0
    q  = q0 - Y^T al,           r0 = H q
    second loop:  be_i = ro_i (y_i . r_i),  r_i = r0 + sum_{j<i} (al_j - be_j) s_j
                  <=>  tril(M^T) w = diag(M) al - H (Y q),     w = al - be
    d  = r0 + S^T w

i.e. four (k x P) matrix-vector products, two k x k triangular solves (fp64), and two more
products to extend M when a pair is stored.  On the GPU that is six launches of csrc/pinn_lbfgs.hip
(`_HipHistory`); `_History` is the same formulation in torch operators (CPU, or no library).  Same arithmetic up to summation order; everything else
(memory update rule, step-size initialisation, strong-Wolfe line search, stopping tests, state
counters) is torch.optim.LBFGS's own: this class subclasses it, re-uses its helpers and its
`_strong_wolfe`, and falls back to its `step` when it cannot apply.
"""
from __future__ import annotations

import torch
from torch.optim import LBFGS as _TorchLBFGS

try:                                     # private helper of torch.optim.lbfgs: same line search, same decisions
    from torch.optim.lbfgs import _strong_wolfe
except Exception:                        # pragma: no cover
    _strong_wolfe = None


class _History:
    """(s, y) pairs as two (m x P) matrices, oldest first, plus M = S Y^T in fp64.  Unused rows are zero
    with a unit diagonal in M, so every product and both triangular solves run over the full m without
    gathering; when the history is full the oldest row is rolled out (24 MB of copies at m = 100,
    P = 30 k: cheaper than the index bookkeeping it replaces)."""

    def __init__(self, m: int, like: torch.Tensor):
        self.m, self.k = m, 0
        self.S = torch.zeros(m, like.numel(), dtype=like.dtype, device=like.device)
        self.Y = torch.zeros_like(self.S)
        self.M = torch.eye(m, dtype=torch.float64, device=like.device)

    def push(self, s: torch.Tensor, y: torch.Tensor):
        if self.k == self.m:
            self.S = torch.roll(self.S, -1, 0)
            self.Y = torch.roll(self.Y, -1, 0)
            self.M = torch.roll(self.M, (-1, -1), (0, 1))
            self.k -= 1
        i = self.k
        self.S[i].copy_(s)
        self.Y[i].copy_(y)
        self.M[i, :] = torch.mv(self.Y, s)      # s_new . y_j   (zero beyond the used rows)
        self.M[:, i] = torch.mv(self.S, y)      # s_i . y_new   (M[i, i] = s . y either way)
        self.k += 1

    def direction(self, g: torch.Tensor, H) -> torch.Tensor:
        """-(inverse-Hessian approximation) g by the recursion written as two triangular solves."""
        q0 = g.neg()
        if self.k == 0:
            return q0 * H
        M = self.M
        b = torch.mv(self.S, q0).double()
        al = torch.linalg.solve_triangular(torch.triu(M), b.unsqueeze(1), upper=True).squeeze(1)
        q = torch.addmv(q0, self.Y.t(), al.to(g.dtype), alpha=-1.0)
        c = torch.mv(self.Y, q).double()
        rhs = torch.diagonal(M) * al - c * H
        w = torch.linalg.solve_triangular(torch.tril(M.t()), rhs.unsqueeze(1), upper=False).squeeze(1)
        return torch.addmv(q * H, self.S.t(), w.to(g.dtype))


class _HipHistory:
    """The same history on the device through libpinn_hip.so (csrc/pinn_lbfgs.hip): S, Y are rings
    (no roll), the recursion is six launches (two row-dot passes, two one-wave triangular solves in
    fp64, two combines) instead of ~25 torch operators."""

    def __init__(self, m: int, like: torch.Tensor):
        from . import _lib
        import ctypes
        self._C, self.lib = ctypes, _lib.load()
        self.m, self.k, self.head, self.P = m, 0, 0, like.numel()
        dev = like.device
        self.S = torch.zeros(m, self.P, dtype=torch.float32, device=dev)
        self.Y = torch.zeros_like(self.S)
        self.M = torch.zeros(m, m, dtype=torch.float64, device=dev)
        self.tmp = torch.zeros(4 * m, dtype=torch.float64, device=dev)
        self.coef = torch.zeros(2 * m, dtype=torch.float32, device=dev)
        self.q = torch.empty(self.P, dtype=torch.float32, device=dev)

    def _p(self, t):
        return self._C.c_void_p(t.data_ptr())

    def _check(self, rc, what):
        if rc != 0:
            from ._lib import PinnError
            raise PinnError(f"{what}: {self.lib.pinn_last_error().decode()}")

    def push(self, s: torch.Tensor, y: torch.Tensor):
        if self.k == self.m:
            slot, self.head = self.head, (self.head + 1) % self.m      # overwrite the oldest pair
        else:
            slot = (self.head + self.k) % self.m
            self.k += 1
        s, y = s.contiguous(), y.contiguous()
        st = self._C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream)
        self._check(self.lib.pinn_lbfgs_push(self._p(self.S), self._p(self.Y), self._p(self.M), self.m, self.P, slot,
                                             self._p(s), self._p(y), st), "pinn_lbfgs_push")

    def direction(self, g: torch.Tensor, H) -> torch.Tensor:
        if self.k == 0:
            return g.neg() * H
        g = g.contiguous()
        d = torch.empty_like(g)
        st = self._C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
        self._check(self.lib.pinn_lbfgs_direction(self._p(self.S), self._p(self.Y), self._p(self.M), self.m, self.P,
                                                  self.head, self.k, self._p(g), float(H), self._p(d), self._p(self.tmp),
                                                  self._p(self.coef), self._p(self.q), st), "pinn_lbfgs_direction")
        return d


def _make_history(m: int, like: torch.Tensor):
    # fp32 parameters on the GPU (the product path): csrc/pinn_lbfgs.hip, and a missing library is an error, not a
    # reason to run something else.  _History (torch operators) serves CPU tensors and float64 — the CPU tests of the
    # optimizer's semantics against torch.optim.LBFGS — and history sizes beyond the kernels' 256 (the reference: 100).
    if like.is_cuda and like.dtype == torch.float32 and m <= 256:
        return _HipHistory(m, like)
    return _History(m, like)


class FlatLBFGS(_TorchLBFGS):
    """torch.optim.LBFGS over ONE flat parameter tensor with the batched recursion above."""

    @torch.no_grad()
    def step(self, closure):
        group = self.param_groups[0]
        if _strong_wolfe is None or len(self._params) != 1 or group["line_search_fn"] not in (None, "strong_wolfe"):
            return super().step(closure)
        closure = torch.enable_grad()(closure)
        lr, max_iter, max_eval = float(group["lr"]), group["max_iter"], group["max_eval"]
        tol_grad, tol_change = group["tolerance_grad"], group["tolerance_change"]
        line_search, m = group["line_search_fn"], group["history_size"]
        state = self.state[self._params[0]]
        state.setdefault("func_evals", 0)
        state.setdefault("n_iter", 0)

        orig_loss = closure()
        loss = float(orig_loss)
        evals = 1
        state["func_evals"] += 1
        g = self._gather_flat_grad()
        if float(g.abs().max()) <= tol_grad:
            return orig_loss

        d, t = state.get("d"), state.get("t")
        hist: _History = state.get("hist")
        H = state.get("H_diag")
        prev_g, prev_loss = state.get("prev_flat_grad"), state.get("prev_loss")

        n_iter = 0
        pending = None      # (y, s, y.s, y.y) of the step just taken, fetched with the stopping tests' values
        while n_iter < max_iter:
            n_iter += 1
            state["n_iter"] += 1
            if state["n_iter"] == 1:
                d = g.neg()
                hist = _make_history(m, g)
                H = 1.0
            else:
                if pending is None:                                       # (first iteration of a later .step call)
                    y, s = g.sub(prev_g), d.mul(t)
                    pending = (y, s) + tuple(torch.stack((y.dot(s), y.dot(y))).tolist())
                y, s, ys, yy = pending
                pending = None
                if ys > 1e-10:
                    hist.push(s, y)
                    H = ys / yy
                d = hist.direction(g, H)
            if prev_g is None:
                prev_g = g.clone(memory_format=torch.contiguous_format)
            else:
                prev_g.copy_(g)
            prev_loss = loss

            gd = g.dot(d)
            gtd_l1 = torch.stack((gd, g.abs().sum())).tolist()           # one synchronisation for both
            t = min(1.0, 1.0 / gtd_l1[1]) * lr if state["n_iter"] == 1 else lr
            if gtd_l1[0] > -tol_change:
                break

            ls_evals = 0
            if line_search is not None:
                x_init = self._clone_param()

                def obj_func(x, step, direction):
                    return self._directional_evaluate(closure, x, step, direction)

                # The line-search budget is whatever the INSTALLED torch.optim.LBFGS.step gives it: torch 2.10
                # (this image) passes max_ls = max_eval - current_evals; older releases used _strong_wolfe's
                # default of 25.  tests/test_lbfgs_cpu.py::test_line_search_budget_is_torchs compares
                # evaluation counts with torch.optim.LBFGS itself, so a torch upgrade that changes the rule
                # shows up there.
                loss, g, t, ls_evals = _strong_wolfe(obj_func, x_init, t, d, loss, g, gd, max_ls=max_eval - evals)
                self._add_grad(t, d)
            else:
                self._add_grad(t, d)
                if n_iter != max_iter:
                    with torch.enable_grad():
                        loss = float(closure())
                    g = self._gather_flat_grad()
                    ls_evals = 1
            evals += ls_evals
            state["func_evals"] += ls_evals

            if n_iter == max_iter or evals >= max_eval:
                break
            # one synchronisation: the stopping tests' maxima and the next memory update's y.s, y.y
            y, s = g.sub(prev_g), d.mul(t)
            gmax, smax, ys, yy = torch.stack((g.abs().max(), s.abs().max(), y.dot(s), y.dot(y))).tolist()
            pending = (y, s, ys, yy)
            if gmax <= tol_grad:
                break
            if smax <= tol_change:
                break
            if abs(loss - prev_loss) < tol_change:
                break

        state.update(d=d, t=t, hist=hist, H_diag=H, prev_flat_grad=prev_g, prev_loss=prev_loss)
        return orig_loss
