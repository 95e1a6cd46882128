"""FlatLBFGS (batched two-loop recursion) against torch.optim.LBFGS itself: same iterates."""
import pytest
import torch

from pinn_depthestimation_amd.lbfgs import FlatLBFGS


def problem(dtype, seed=0, n=64, d=6, h=8):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, d, generator=g, dtype=dtype)
    T = torch.sin(X.sum(1, keepdim=True))
    P = d * h + h + h + 1
    theta0 = 0.3 * torch.randn(P, generator=g, dtype=dtype)

    def loss_of(theta):
        W1 = theta[:d * h].view(d, h); b1 = theta[d * h:d * h + h]
        W2 = theta[d * h + h:d * h + 2 * h].view(h, 1); b2 = theta[-1]
        return ((torch.tanh(X @ W1 + b1) @ W2 + b2 - T) ** 2).mean()
    return theta0, loss_of


def run(opt_cls, dtype, hist, line_search, max_iter):
    theta0, loss_of = problem(dtype)
    theta = torch.nn.Parameter(theta0.clone())
    opt = opt_cls([theta], lr=1.0, max_iter=max_iter, max_eval=None, history_size=hist, tolerance_grad=1e-12,
                  tolerance_change=1e-14, line_search_fn=line_search)
    trace = []

    def closure():
        opt.zero_grad()
        loss = loss_of(theta)
        loss.backward()
        trace.append(float(loss))
        return loss
    opt.step(closure)
    st = opt.state[theta]
    return theta.detach().clone(), trace, st["n_iter"], st["func_evals"]


@pytest.mark.parametrize("hist", [3, 100])
@pytest.mark.parametrize("line_search", ["strong_wolfe", None])
def test_matches_torch_lbfgs_fp64(hist, line_search):
    it = 40 if line_search else 15
    a = run(torch.optim.LBFGS, torch.float64, hist, line_search, it)
    b = run(FlatLBFGS, torch.float64, hist, line_search, it)
    assert a[2] == b[2] and a[3] == b[3]                       # same iteration and evaluation counts
    n = min(len(a[1]), len(b[1]))
    ta, tb = torch.tensor(a[1][:n]), torch.tensor(b[1][:n])
    assert torch.allclose(ta, tb, rtol=1e-7, atol=1e-12), (ta - tb).abs().max()
    assert (a[0] - b[0]).abs().max() < 1e-6 * a[0].abs().max()


def test_fp32_descent_and_second_step_continues_history():
    theta0, loss_of = problem(torch.float32)
    theta = torch.nn.Parameter(theta0.clone())
    opt = FlatLBFGS([theta], lr=1.0, max_iter=10, history_size=5, line_search_fn="strong_wolfe")
    losses = []

    def closure():
        opt.zero_grad()
        loss = loss_of(theta)
        loss.backward()
        losses.append(float(loss))
        return loss
    opt.step(closure)
    first = losses[-1]
    opt.step(closure)                                         # state (history, d, t) carried over, as in torch
    assert losses[-1] < first < losses[0]
    assert opt.state[theta]["n_iter"] == 20


def _stiff_run(opt_cls, max_iter, max_eval):
    """sum sqrt(1 + (a_i x_i)^2): asymptotically linear, curvatures 1 : 900 — quasi-Newton steps overshoot
    and single line searches take several evaluations (more than a small max_eval leaves)."""
    a = torch.tensor([1.0, 30.0, 0.2, 5.0], dtype=torch.float64)
    x = torch.nn.Parameter(torch.tensor([3.0, -2.0, 40.0, 1.0], dtype=torch.float64))
    opt = opt_cls([x], lr=1.0, max_iter=max_iter, max_eval=max_eval, history_size=10, tolerance_grad=1e-14,
                  tolerance_change=1e-16, line_search_fn="strong_wolfe")
    trace = []

    def closure():
        opt.zero_grad()
        loss = torch.sqrt(1 + (a * x) ** 2).sum()
        loss.backward()
        trace.append(float(loss.detach()))
        return loss
    opt.step(closure)
    return x.detach().clone(), trace, opt.state[x]["func_evals"]


@pytest.mark.parametrize("max_iter,max_eval", [(2, 3), (5, 6), (10, 12), (40, None)])
def test_line_search_budget_is_torchs(max_iter, max_eval):
    """The per-search evaluation budget must be the installed torch.optim.LBFGS's (torch 2.10: max_ls =
    max_eval - evaluations so far; releases before it: 25).  In the first two cases a single line search
    wants more evaluations than max_eval leaves, so any other rule changes the evaluation count."""
    a = _stiff_run(torch.optim.LBFGS, max_iter, max_eval)
    b = _stiff_run(FlatLBFGS, max_iter, max_eval)
    assert a[2] == b[2], (a[2], b[2])
    assert len(a[1]) == len(b[1])
    assert torch.allclose(torch.tensor(a[1]), torch.tensor(b[1]), rtol=1e-9, atol=1e-14)
    assert (a[0] - b[0]).abs().max() < 1e-9
