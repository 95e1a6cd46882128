"""pinn_loss_grad_adam_step: train.py:189-193 (loss_func + backward + Adam.step) in two launches.  The folded kernel
must give what the separate calls give — the same partial sums in the same order and pinn_adam_step's arithmetic — so
everything is compared BIT FOR BIT with the loss call followed by pinn_adam_step, over several iterations (the
second and later ones run on the packed weights the first one left in the workspace).  Bit for bit where the pass itself
is reproducible (the cooperative kernel: 8 x 64 at small N); the tile kernel accumulates a workgroup's gradient in lock
order (run-to-run differences of ~1e-6), so there the two runs are compared as two runs of the classic path
would be."""
import numpy as np
import pytest
import torch

from oracle import pinn_oracle as O          # parameter initialisation only
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd._lib import ENGINE_FUSED_COOP, ENGINE_FUSED_TILE, ENGINE_WIDE
from pinn_depthestimation_amd.trainer import PINN

pytestmark = pytest.mark.gpu

CASES = {
    # name: d_in, d_out, L, W, grad_cols, residual, inputs, outputs, fid cols, N_res, N_fid (None: newmethod — one point set)
    "ns8x64_res_only": (3, 4, 8, 64, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v"), (), 243, 0),
    "ns8x64_split": (3, 4, 8, 64, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v"), (2, 3), 243, 12),
    "pe10x10_split": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"), (0, 1, 2, 3, 4, 5), 243, 12),
    "co100x20_newmethod": (2, 3, 100, 20, (0, 1), "continuity_only", ("x", "y"), ("U", "V", "h"), (0, 1), 1251, None),
    # enough points for AUTO to pick the batch kernel (k_fused_batch: k-step-major packing, its own gradient sinks):
    # the folded finish kernel must un-permute / re-pack in that layout too
    "pe10x10_split_batch": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"), (0, 1, 2, 3, 4, 5), 6000, 12),
    "co40x20_newmethod_batch": (2, 3, 40, 20, (0, 1), "continuity_only", ("x", "y"), ("U", "V", "h"), (0, 1), 5000, None),
}


def _setup(name, engine=0):
    d_in, d_out, L, W, gc, res, inn, outn, fid, n_res, n_fid = CASES[name]
    g = torch.Generator().manual_seed(77)
    params = O.init_params(O.layer_sizes(d_in, L, W, d_out), "xavier", g)
    if res == "physics_equation":
        params[-1][outn.index("h")] = 0.75; params[-1][outn.index("eta_mean")] = 0.0
    desc = NetDesc(d_in, d_out, L, W, gc, engine=engine)
    spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
    nf = n_res if n_fid is None else n_fid
    X = (torch.rand(n_res + (0 if n_fid is None else n_fid), d_in, generator=g) * 2 - 1).cuda()
    T = torch.rand(nf, len(fid), generator=g).cuda() if fid else None
    scale = torch.full((spec.n_terms,), 1.0 / n_res).cuda()
    cscale = torch.full((len(fid),), 1.0 / max(nf, 1)).cuda() if fid else None
    return desc, spec, O.flatten(params).cuda(), X, T, scale, cscale, list(fid), (-1 if n_fid is None else n_res)


@pytest.mark.parametrize("name", list(CASES))
def test_folded_iteration_is_bitwise_the_loss_call_plus_adam_step(name):
    desc, spec, flat0, X, T, scale, cscale, fid, n_res = _setup(name)
    P = flat0.numel()
    runs = {}
    for mode in ("classic", "folded"):
        eng = Engine(desc)
        th, m, v, grad = flat0.clone(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
        ts, cs = torch.zeros(spec.n_terms, device="cuda"), torch.zeros(max(len(fid), 1), device="cuda")[:len(fid)]
        hist = []
        for step in range(1, 7):
            lr = 1e-3 * 0.8 ** (step // 3)
            if step == 4:
                th.mul_(1.0 + 1e-3)        # a write by torch between iterations: the packed copy must not be trusted
            if mode == "folded":
                assert eng.loss_grad_adam_step(spec, scale, th, X, n_res if n_res >= 0 else -1, grad, m, v, step, lr, T=T,
                                               out_col=fid, col_scale=cscale, term_sums=ts, col_sums=cs if fid else None)
            else:
                grad.zero_()
                if not fid:
                    eng.residual_loss_grad(spec, scale, th, X, grad, sums=ts)
                elif n_res < 0:
                    eng.residual_mse_loss_grad(spec, scale, T, fid, cscale, th, X, grad, term_sums=ts, col_sums=cs)
                else:
                    eng.residual_mse_split_loss_grad(spec, scale, T, fid, cscale, th, X, n_res, grad, term_sums=ts, col_sums=cs)
                eng.adam_step(th, grad, m, v, step, lr)
            hist.append((ts.clone(), cs.clone(), grad.clone()))
        torch.cuda.synchronize()
        runs[mode] = (th, m, v, hist)
    a, b = runs["classic"], runs["folded"]
    exact = name.startswith("ns8x64")          # cooperative kernel: reproducible pass
    same = torch.equal if exact else (lambda x, y: bool(torch.allclose(x, y, rtol=2e-4, atol=1e-7 * float(x.abs().max()))))
    for k, (ha, hb) in enumerate(zip(a[3], b[3])):
        assert torch.equal(ha[0], hb[0]) and torch.equal(ha[1], hb[1]) if exact else \
            torch.allclose(ha[0], hb[0], rtol=1e-5) and torch.allclose(ha[1], hb[1], rtol=1e-5), f"loss sums differ at iteration {k + 1}"
        rel = float((ha[2] - hb[2]).norm() / ha[2].norm())
        assert rel == 0.0 if exact else rel < 5e-6, f"gradient differs at iteration {k + 1}: {rel:.2e}"
    # Adam divides by sqrt(v): a 1e-6 relative difference of a gradient entry moves its parameter by ~lr * 1e-6
    assert same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2])
    assert bool(torch.isfinite(b[0]).all()) and float((b[0] - flat0).abs().max()) > 0


@pytest.mark.parametrize("engine", [ENGINE_FUSED_TILE, ENGINE_FUSED_COOP])
def test_folded_iteration_on_both_fused_kernels(engine):
    """The residual-only request is one pass on either kernel; classic and folded agree bit for bit on each."""
    desc, spec, flat0, X, T, scale, cscale, fid, n_res = _setup("ns8x64_res_only", engine)
    P = flat0.numel()
    out = []
    for folded in (False, True):
        eng = Engine(desc)
        th, m, v, grad = flat0.clone(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
        ts = torch.zeros(spec.n_terms, device="cuda")
        for step in (1, 2, 3):
            if folded:
                assert eng.loss_grad_adam_step(spec, scale, th, X, X.shape[0], grad, m, v, step, 1e-3, term_sums=ts)
            else:
                grad.zero_(); eng.residual_loss_grad(spec, scale, th, X, grad, sums=ts); eng.adam_step(th, grad, m, v, step, 1e-3)
        out.append((th, ts.clone()))
    if engine == ENGINE_FUSED_COOP:
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    else:                                        # tile kernel: lock-ordered accumulation, see the module docstring
        assert torch.allclose(out[0][0], out[1][0], rtol=2e-4, atol=1e-7) and torch.allclose(out[0][1], out[1][1], rtol=1e-5)


def test_weighted_losses_from_the_finishing_kernel():
    desc, spec, flat0, X, T, scale, cscale, fid, n_res = _setup("ns8x64_split")
    P = flat0.numel()
    eng = Engine(desc)
    th, m, v, grad = flat0.clone(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    ts, cs = torch.zeros(spec.n_terms, device="cuda"), torch.zeros(len(fid), device="cuda")
    rows = torch.rand(3, len(fid) + spec.n_terms, generator=torch.Generator().manual_seed(2)).cuda()
    out = torch.zeros(3, device="cuda")
    assert eng.loss_grad_adam_step(spec, scale, th, X, n_res, grad, m, v, 1, 1e-3, T=T, out_col=fid, col_scale=cscale,
                                   term_sums=ts, col_sums=cs, loss_rows=rows, losses=out)
    ref = rows.double() @ torch.cat([cs, ts]).double()
    assert torch.allclose(out.double(), ref, rtol=1e-6)


def test_requests_that_are_not_one_fused_pass_are_refused_without_side_effects():
    # the wide engine, and a split request too large for the cooperative kernel (it runs as two passes)
    for name, desc_kw, N in (("wide", dict(d_in=3, d_out=4, L=3, W=128), 300), ("big_split", dict(d_in=3, d_out=4, L=8, W=64), 20000)):
        desc = NetDesc(desc_kw["d_in"], desc_kw["d_out"], desc_kw["L"], desc_kw["W"], (0, 1, 2), engine=ENGINE_WIDE if name == "wide" else 0)
        spec = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), desc.grad_cols, ("h", "z", "u", "v"))
        eng = Engine(desc)
        g = torch.Generator().manual_seed(3)
        th = torch.randn(desc.n_params, generator=g).mul_(0.1).cuda()
        X = (torch.rand(N + 12, 3, generator=g) * 2 - 1).cuda()
        T = torch.rand(12, 2, generator=g).cuda()
        before = th.clone()
        m, v, grad = (torch.zeros_like(th) for _ in range(3))
        ts, cs = torch.zeros(spec.n_terms, device="cuda"), torch.zeros(2, device="cuda")
        ok = eng.loss_grad_adam_step(spec, torch.ones(spec.n_terms, device="cuda"), th, X, N, grad, m, v, 1, 1e-3, T=T,
                                     out_col=[2, 3], col_scale=torch.ones(2, device="cuda"), term_sums=ts, col_sums=cs)
        torch.cuda.synchronize()
        assert ok is False, name
        assert torch.equal(th, before) and float(m.abs().max()) == 0.0 and float(grad.abs().max()) == 0.0


def _cfg(adam_it, fid_outputs):
    return {
        "layers": {"input_features": 3, "hidden_layers": 8, "hidden_width": 64, "output_features": 4},
        "adam_optimizer": {"max_it": adam_it, "learning_rate": 1e-3, "scheduler_step_size": 7, "scheduler_gamma": 0.8},
        "lbfgs_optimizer": {"max_it": 0, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                            "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
        "loss": {"weight_fid_loss": 1, "weight_res_loss": 1, **{f"weight_{k}_loss": 1.0 for k in fid_outputs}},
        "data_fidelity": {"inputs": ["t", "x", "y"], "outputs": list(fid_outputs)},
        "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "txy"}, "outputs": ["h", "z", "u", "v"]},
    }


@pytest.mark.parametrize("fid_outputs", [(), ("u", "v")])
def test_trainer_with_and_without_the_folded_update_walks_the_same_trajectory(tmp_path, fid_outputs):
    """30 Adam iterations with StepLR steps, a checkpoint inside the run (that iteration takes the classic path: the
    reference saves the pre-update weights from inside loss_func) and per-iteration logging: identical parameters and
    the same logged losses (to fp32 rounding of the final weighted sum) with fold_adam on and off."""
    rs = np.random.RandomState(5)
    Xr = rs.rand(243, 3).astype(np.float32) * 2 - 1
    Xf = rs.rand(12, 3).astype(np.float32) * 2 - 1 if fid_outputs else None
    Tf = rs.rand(12, len(fid_outputs)).astype(np.float32) if fid_outputs else None
    res = {}
    for fold in (False, True):
        torch.manual_seed(1234)
        tr = PINN(Xf, Tf, Xr, _cfg(30, fid_outputs), log_every=1, checkpoint_every=10, log_dir=str(tmp_path / f"f{int(fold)}"),
                  fold_adam=fold)
        if fold:
            tr.train_adam(30)                         # runs of 9 iterations per call between the checkpoints
        else:
            for _ in range(30):
                tr.adam_step()
        assert tr._folded_iters == (27 if fold else 0)    # iterations 10, 20, 30 save a checkpoint: classic path
        assert tr.iter == 30 and tr._adam_step == 30
        res[fold] = (tr.dnn.flat_params().clone(), list(tr.history), torch.load(str(tmp_path / f"f{int(fold)}" / "model_20.state.pth"), weights_only=True))
    assert torch.equal(res[False][0], res[True][0])
    # the logged losses: the folded kernel forms them in double from the same sums, the classic path by an fp32 mat-vec
    assert len(res[True][1]) == 30 and [r[0] for r in res[True][1]] == [r[0] for r in res[False][1]]
    assert np.allclose(np.array(res[True][1])[:, 1:], np.array(res[False][1])[:, 1:], rtol=2e-6, atol=0.0)
    for k in res[False][2]:
        assert torch.equal(res[False][2][k], res[True][2][k]), k


def test_packed_weights_are_not_trusted_after_another_call_on_the_engine():
    """Between two folded iterations a forward pass with OTHER parameters re-packs the workspace: the next folded
    iteration must pack again (it equals the classic path bit for bit only if it does)."""
    desc, spec, flat0, X, T, scale, cscale, fid, n_res = _setup("ns8x64_res_only")
    P = flat0.numel()
    out = []
    for folded in (False, True):
        eng = Engine(desc)
        th, m, v, grad = flat0.clone(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
        ts = torch.zeros(spec.n_terms, device="cuda")
        for step in (1, 2, 3):
            if folded:
                assert eng.loss_grad_adam_step(spec, scale, th, X, X.shape[0], grad, m, v, step, 1e-3, term_sums=ts)
                eng.forward(torch.zeros_like(th), X)            # someone else's parameters through the same engine
            else:
                grad.zero_(); eng.residual_loss_grad(spec, scale, th, X, grad, sums=ts); eng.adam_step(th, grad, m, v, step, 1e-3)
        out.append(th)
    assert torch.equal(out[0], out[1])


def test_runs_of_iterations_log_the_same_iterations_as_the_per_iteration_loop(tmp_path):
    """log_every = 7 with a 16-row log ring and no checkpoints: train_adam enqueues runs that stop when the ring is full;
    the logged iterations and their losses equal those of the classic per-iteration loop."""
    rs = np.random.RandomState(6)
    Xr = rs.rand(243, 3).astype(np.float32) * 2 - 1
    res = {}
    for fold in (False, True):
        torch.manual_seed(99)
        tr = PINN(None, None, Xr, _cfg(120, ()), log_every=7, checkpoint_every=0, log_flush_every=16, fold_adam=fold)
        if fold:
            tr.train_adam(120)
        else:
            for _ in range(120):
                tr.adam_step()
        res[fold] = (tr.dnn.flat_params().clone(), list(tr.history), tr._folded_iters)
    assert res[True][2] == 120 and res[False][2] == 0
    assert torch.equal(res[False][0], res[True][0])
    assert [r[0] for r in res[True][1]] == [r[0] for r in res[False][1]] == [i for i in range(1, 121) if i % 7 == 0]
    assert np.allclose(np.array(res[True][1])[:, 1:], np.array(res[False][1])[:, 1:], rtol=2e-6, atol=0.0)


def test_writes_through_the_modules_parameters_invalidate_the_packed_weights():
    """ADVICE r2 (engine.py packed-weights token): DNN's Parameters alias the flat buffer through `.data`, so torch bumps
    THEIR version counters, not the flat tensor's.  A write through a Parameter (load_state_dict, p.mul_()) between two
    folded iterations must make the next pass re-pack; the token now carries DNN.write_token()."""
    from pinn_depthestimation_amd.dnn import DNN
    desc, spec, flat0, X, T, scale, cscale, fid, n_res = _setup("ns8x64_res_only")
    P = flat0.numel()
    g = torch.Generator().manual_seed(5)
    other = O.flatten(O.init_params(O.layer_sizes(3, 8, 64, 4), "xavier", g))
    out = {}
    for mode in ("classic", "folded"):
        torch.manual_seed(0)
        dnn = DNN([3] + [64] * 8 + [4], 0.0, "xavier").cuda()
        th = dnn.flat_params()
        th.copy_(flat0)
        donor = DNN([3] + [64] * 8 + [4], 0.0, "xavier")
        donor.flat_params().copy_(other)
        eng = Engine(desc)
        m, v, grad = torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
        ts = torch.zeros(spec.n_terms, device="cuda")
        sums = []
        for step in range(1, 6):
            if step == 3:
                v_flat = th._version
                dnn.load_state_dict(donor.state_dict())          # param.copy_(): bumps the Parameters' counters only
                assert th._version == v_flat, "the premise of this test: the flat buffer's counter does not see it"
                assert torch.equal(dnn.flat_params().cpu(), other)
            if step == 5:
                with torch.no_grad():
                    next(dnn.parameters()).mul_(1.5)             # an in-place write through one Parameter
            if mode == "folded":
                assert eng.loss_grad_adam_step(spec, scale, th, X, X.shape[0], grad, m, v, step, 1e-3, term_sums=ts,
                                               params_token=dnn.write_token())
            else:
                grad.zero_(); eng.residual_loss_grad(spec, scale, th, X, grad, sums=ts); eng.adam_step(th, grad, m, v, step, 1e-3)
            sums.append(ts.clone())
        torch.cuda.synchronize()
        out[mode] = (th.clone(), sums)
    for k, (a, b) in enumerate(zip(out["classic"][1], out["folded"][1])):
        assert torch.equal(a, b), f"loss sums differ at iteration {k + 1}: the pass ran on stale packed weights"
    assert torch.equal(out["classic"][0], out["folded"][0])


def test_loss_rows_keep_their_column_stride_when_no_fidelity_point_is_given():
    """n_res == N with n_cols > 0 runs as a residual-only pass, but loss_rows is (rows, n_cols + n_terms) by
    the header's contract: the column sums are zeros, the term sums sit BEHIND them (they used to be read with stride
    n_terms and multiplied by the column weights)."""
    desc, spec, flat0, X, T, scale, cscale, fid, n_res = _setup("ns8x64_split")
    P, N = flat0.numel(), X.shape[0]
    eng = Engine(desc)
    th, m, v, grad = flat0.clone(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    ts, cs = torch.zeros(spec.n_terms, device="cuda"), torch.full((len(fid),), 7.0, device="cuda")
    rows = torch.rand(3, len(fid) + spec.n_terms, generator=torch.Generator().manual_seed(2)).cuda()
    out = torch.zeros(3, device="cuda")
    full_scale = torch.full((spec.n_terms,), 1.0 / N, device="cuda")
    assert eng.loss_grad_adam_step(spec, full_scale, th, X, N, grad, m, v, 1, 1e-3, T=None, out_col=fid, col_scale=cscale,
                                   term_sums=ts, col_sums=cs, loss_rows=rows, losses=out)
    assert float(cs.abs().sum()) == 0.0                       # no fidelity point: zero column sums
    ref = rows[:, len(fid):].double() @ ts.double()
    assert torch.allclose(out.double(), ref, rtol=1e-6), (out, ref)
