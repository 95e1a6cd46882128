"""GPU tests of the rows either side of the hot path: grid inference with the
physics-only L-BFGS fine-tune (test.py), the SciPy L-BFGS-B stage (BASELINE configs[4]), the
.mat prediction dump (train_newmethod.py:141-153) and checkpoint round trips."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

from oracle import pinn_oracle as O  # noqa: E402
from tests.golden_util import CMB, layers_of, load, ns_config, rel_l2, state_dict  # noqa: E402

pytestmark = pytest.mark.gpu


def test_grid_inference_and_physics_only_finetune():
    """test.py on config_CMB.json's 81 x 261 grid (21 141 points), 2 -> 8x64 -> 6, physics_equation."""
    import dnn
    from pinn_depthestimation_amd.inference import Tester
    z = load("g4_pe_8x64_conditioned.npz")
    sd = state_dict(z)
    cfg = dict(CMB)
    cfg["layers"] = dict(CMB["layers"], hidden_layers=8, hidden_width=64)
    cfg["data_test"] = {"inputs": CMB["data_residual"]["inputs"], "outputs": CMB["data_residual"]["outputs"],
                        "nx": 81, "ny": 261, "x_min": 25.0, "x_max": 33.0, "y_min": -13.0, "y_max": 13.0}
    model = dnn.DNN(layers_of(sd), 0.0, "xavier")
    model.load_state_dict(sd)
    xs, ys = np.meshgrid(np.linspace(-1, 1, 81), np.linspace(-1, 1, 261))
    grid = np.hstack([xs.reshape(-1, 1), ys.reshape(-1, 1)]).astype(np.float32)
    t = Tester(model, cfg)
    pred0 = t.test(grid, input_min_max={"x": (25.0, 33.0), "y": (-13.0, 13.0)}, perform_optimization=False)
    params = O.params_from_state_dict(sd)
    ref0 = O.mlp_forward([p.double() for p in params], torch.from_numpy(grid).double()).numpy()
    assert pred0.shape == (21141, 6) and np.abs(pred0 - ref0).max() < 2e-6
    assert t.plot_pred_h.shape == (261, 81) and np.allclose(t.plot_input_x[0, :], np.linspace(25, 33, 81), atol=1e-5)
    # one LBFGS.step (max_iter=1, max_eval=2, history 10) on the residual alone, test.py:44-54,92-104
    pred1 = t.test(grid, perform_optimization=True)
    Xg = torch.from_numpy(grid)
    loss_fn = lambda p: O.residual_loss(p, Xg, "physics_equation", [0, 1], [0, 1, 2, 3, 4, 5], (0, 1))
    losses, p_end = O.lbfgs_trajectory(params, loss_fn, max_iter=1, max_eval=2, history_size=10)
    ref1 = O.mlp_forward(p_end, Xg).numpy()
    assert abs(float(t.last_loss) - losses[-1]) / losses[-1] < 1e-3
    assert np.abs(pred1 - ref1).max() < 1e-4 and np.abs(pred1 - pred0).max() > 1e-6


def test_scipy_lbfgsb_stage_after_adam_warm_start():
    """BASELINE configs[4]: full-batch closure under scipy.optimize L-BFGS-B after the Adam stage."""
    import dnn
    from pinn_depthestimation_amd.lbfgsb import LBFGSBOptimizer
    from pinn_depthestimation_amd.trainer import pinn
    z7, z8 = load("g7_adam_ns_8x64.npz"), load("g8_lbfgs_ns_8x64.npz")
    model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict(state_dict(z7, "sd_end/"))
    tr = pinn(None, None, z7["X"][:2000], ns_config(0), dnn=model, log_every=10 ** 9, checkpoint_every=0)
    opt = LBFGSBOptimizer(tr, {"maxiter": 40, "maxfun": 60})
    res = opt.minimize()
    assert abs(opt.losses[0] - z8["losses"][0]) / z8["losses"][0] < 5e-6      # same start as the torch-LBFGS golden
    assert res.fun < 0.1 * opt.losses[0] and res.nit >= 10
    assert abs(float(tr.loss_func()) - res.fun) / res.fun < 1e-4             # best point left in the network


def test_mat_dump_and_checkpoint_formats(tmp_path):
    import dnn
    from scipy.io import loadmat
    from pinn_depthestimation_amd.trainer import pinn
    z = load("g9_newmethod_at50k.npz")
    sd = state_dict(z, "8x64/sd/")
    cfg = {"layers": {"input_features": 2, "hidden_layers": 8, "hidden_width": 64, "output_features": 3,
                      "dropout_rate": 0.0, "init_type": "xavier"},
           "adam_optimizer": {"max_it": 3, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
           "lbfgs_optimizer": {"max_it": 0}, "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
           "data": {"inputs": {"x": {"requires_grad": ["true"]}, "y": {"requires_grad": ["true"]}},
                    "trues": ["U", "V"], "unknowns": ["h"]}}
    model = dnn.DNN(layers_of(sd), 0.0, "xavier")
    model.load_state_dict(sd)
    T = np.concatenate([z["U"], z["V"]], 1)
    mat = str(tmp_path / "data_at2.mat")
    tr = pinn(z["X"], T, z["X"], cfg, dnn=model, log_dir=str(tmp_path), log_every=1, checkpoint_every=2,
              mat_dump_iter=2, mat_dump_path=mat)
    tr.train()
    m = loadmat(mat)
    for k in ("pred_U", "pred_V", "pred_h"):                                  # data_at50k.mat's variables
        assert m[k].shape == (12514, 1) and m[k].dtype == np.float32
    whole = torch.load(tmp_path / "model_2.pth", weights_only=False)          # torch.save(self.dnn), train.py:179
    sd2 = torch.load(tmp_path / "model_2.state.pth", weights_only=True)
    assert type(whole).__name__ == "DNN" and list(sd2) == list(sd)
    assert torch.equal(whole.state_dict()["layers.layer_3.weight"].cpu(), sd2["layers.layer_3.weight"].cpu())
    rows = open(tmp_path / "log.txt").read().splitlines()
    assert rows[0].startswith("Epoch, Fidelity Loss") and len(rows) == 4 and rows[3].startswith("3, ")


def test_minibatch_resampled_collocation_is_seeded_and_descends():
    """resampled collocation mini-batches (the reference is full-batch only)."""
    import dnn
    from pinn_depthestimation_amd.trainer import pinn
    z0, z7 = load("g1_g3_ns_8x64.npz"), load("g7_adam_ns_8x64.npz")
    runs = []
    for _ in range(2):
        model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
        model.load_state_dict(state_dict(z0))
        tr = pinn(None, None, z7["X"], ns_config(60), dnn=model, log_every=1, checkpoint_every=0,
                  residual_batch=2048, seed=77)
        tr.train()
        runs.append(np.array([h[3] for h in tr.history]))
    # same seed -> same batches; the per-workgroup LDS accumulation order is not fixed, so two runs
    # agree to rounding (observed ~1e-8), not bit for bit
    assert np.allclose(runs[0], runs[1], rtol=1e-5, atol=0)
    assert runs[0][-5:].mean() < 0.05 * runs[0][:5].mean()       # and it trains


def test_minibatch_with_small_fidelity_set_resamples_every_iteration():
    """config_CMB-like: 12 fidelity points (merged into the collocation launch, trainer.HipEvaluator) PLUS a
    resampled collocation mini-batch.  The merged [collocation ; fidelity] matrix must follow the batch of the
    CURRENT iteration: with it keyed on data_ptr() the caching allocator's block re-use froze the first batch.
    Checked three ways: the merged path gives the same losses as the two-launch path (merge_sets=False) for
    the same seed; the residual term changes from one iteration to the next by what resampling does at FIXED
    parameters (no Adam step in between); and a run whose generator is re-seeded each iteration (a frozen
    batch on purpose) differs from it."""
    import dnn
    from pinn_depthestimation_amd.trainer import pinn
    z0, z7 = load("g1_g3_ns_8x64.npz"), load("g7_adam_ns_8x64.npz")
    rng = np.random.RandomState(5)
    Xf = rng.uniform(-1, 1, (12, 3)).astype(np.float32)
    Tf = rng.uniform(-0.2, 0.8, (12, 4)).astype(np.float32)
    cfg = ns_config(0)
    cfg["data_fidelity"] = {"inputs": ["t", "x", "y"], "outputs": ["h", "z", "u", "v"]}
    cfg["loss"] = {"weight_fid_loss": 1, "weight_res_loss": 1, "weight_h_loss": 1, "weight_z_loss": 1,
                   "weight_u_loss": 1, "weight_v_loss": 1}

    def make(merge, frozen=False):
        model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
        model.load_state_dict(state_dict(z0))
        tr = pinn(Xf, Tf, z7["X"], cfg, dnn=model, log_every=1, checkpoint_every=0, residual_batch=256, seed=9)
        tr.evaluator.merge_sets = merge
        out = []
        for _ in range(6):                       # loss_func only: parameters stay fixed, only the batch moves
            if frozen:
                tr._gen.manual_seed(9)
            tr.loss_func()
            out.append([t.item() for t in tr.last])
        return np.array(out)
    merged, split, frozen = make(True), make(False), make(True, frozen=True)
    assert np.allclose(merged, split, rtol=2e-5), np.abs(merged - split).max()
    assert np.allclose(merged[:, 0], merged[0, 0], rtol=1e-6)                 # fidelity term: same 12 points
    res = merged[:, 1]
    assert len(set(np.round(res / res[0], 4))) == len(res)                     # six different batches
    assert np.allclose(frozen[:, 1], frozen[0, 1], rtol=1e-6)                  # the control really is frozen
    assert not np.allclose(res[1:], frozen[1:, 1], rtol=1e-3)


def test_corrected_radiation_stress_switch():
    """corrected=True is an extension: E = 1/8*rho*g*Hrms^2 instead of the reference's
    exact zero.  Checked against the same formulas under fp64 autograd; Hrms and k now get gradient."""
    import dnn
    import physics
    z = load("g4_pe_8x64_conditioned.npz")
    sd = state_dict(z)
    with torch.no_grad():
        sd["layers.layer_8.bias"][4] = 0.2     # Hrms ~ 0.2 m
        sd["layers.layer_8.bias"][5] = 1.0     # k ~ 1 rad/m  (2kh ~ 1.5: sinh well conditioned)
    model = dnn.DNN(layers_of(sd), 0.0, "xavier")
    model.load_state_dict(sd)
    model.to("cuda")
    X = z["X"]
    c = [torch.tensor(X[:, i:i + 1].astype(np.float64), requires_grad=True).float().cuda() for i in range(2)]
    pred = model(torch.cat(c, dim=-1))
    cols = [pred[:, i:i + 1] for i in range(6)]
    loss = physics.physics_equation(c[0], c[1], *cols, corrected=True)
    model.zero_grad()
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
    # fp64 autograd restatement of the corrected formulas
    p64 = [q.double().requires_grad_(True) for q in O.params_from_state_dict(sd)]
    xc = O.split_columns(torch.from_numpy(X).double(), (0, 1))
    Y = O.mlp_forward(p64, torch.cat(xc, -1))
    h, U, V, eta, Hrms, k = [Y[:, i:i + 1] for i in range(6)]
    d = O.compute_gradient
    g_, rho, cd = 9.81, 1025, 0.002
    inv = 1 / (rho * (eta + h))
    E = 0.125 * rho * g_ * Hrms ** 2
    ratio = k * h / torch.sinh(2 * k * h)
    fc = d(U, xc[0]) + d(V, xc[1])
    fx = U * d(U, xc[0]) + V * d(U, xc[1]) + g_ * d(eta, xc[0]) + inv * (rho * cd * U * abs(U)) + inv * d(E * (2 * ratio + 0.5), xc[0])
    fy = U * d(V, xc[0]) + V * d(V, xc[1]) + g_ * d(eta, xc[1]) + inv * (rho * cd * V * abs(V)) + inv * d(E * ratio, xc[1])
    ref = (fc ** 2).mean() + (fx ** 2).mean() + (fy ** 2).mean()
    gref = O.flat_grad(ref, p64)
    assert abs(loss.item() - float(ref)) / float(ref) < 1e-4
    assert rel_l2(got, gref) < 1e-3
    W = layers_of(sd)[-2]
    last_w = got[got.numel() - 6 * W - 6: got.numel() - 6].reshape(6, W)
    assert torch.count_nonzero(last_w[4:6]) > 0          # Hrms, k participate now


def _shape_cfg(d_in, hidden, width, d_out, inputs, grad, outputs, variant="train"):
    rg = lambda k: {"requires_grad": ["true" if k in grad else "false"]}
    cfg = {"layers": {"input_features": d_in, "hidden_layers": hidden, "hidden_width": width, "output_features": d_out},
           "adam_optimizer": {"max_it": 3, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
           "lbfgs_optimizer": {"max_it": 3.0, "learning_rate": 1, "max_evaluation": 6.0, "history_size": 100,
                               "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
           "loss": {"weight_fid_loss": 1, "weight_res_loss": 1}}
    if variant == "newmethod":
        cfg["data"] = {"inputs": {k: rg(k) for k in inputs}, "trues": outputs[:-1], "unknowns": outputs[-1:]}
    else:
        cfg["data_fidelity"] = {"inputs": list(inputs), "outputs": list(outputs)}
        cfg["data_residual"] = {"inputs": {k: rg(k) for k in inputs}, "outputs": {k: {"file": k} for k in outputs}}
    return cfg


@pytest.mark.parametrize("name,cfg", [
    ("config_CMB.json", _shape_cfg(2, 10, 10, 6, "xy", "xy", ["h", "U", "V", "eta_mean", "Hrms", "k"])),
    ("config_CMB_h.json", _shape_cfg(2, 100, 20, 3, "xy", "xy", ["U", "V", "h"], "newmethod")),
    ("config.json", _shape_cfg(5, 100, 20, 4, "txyuv", "txy", ["h", "z", "u", "v"])),
    ("config_txyz.json", _shape_cfg(4, 20, 20, 4, "txyz", "txy", ["h", "z", "u", "v"])),
])
def test_every_reference_config_shape_trains(name, cfg):
    """The four network/variable layouts the reference's JSON files describe (train.py:52-56,86-88),
    on synthetic data: Adam steps then one LBFGS.step run end to end and the loss stays finite and drops."""
    from pinn_depthestimation_amd.trainer import pinn
    torch.manual_seed(3)
    g = np.random.RandomState(3)
    d_in, d_out = cfg["layers"]["input_features"], cfg["layers"]["output_features"]
    nf = d_out - 1 if "data" in cfg else d_out
    Xr = g.uniform(-1, 1, (500, d_in)).astype(np.float32)
    Xf = Xr if "data" in cfg else g.uniform(-1, 1, (40, d_in)).astype(np.float32)
    Tf = g.uniform(0.2, 0.8, (Xf.shape[0], nf)).astype(np.float32)
    tr = pinn(Xf, Tf, Xr, cfg, log_every=1, checkpoint_every=0)
    if tr.spec.name == "physics_equation":
        with torch.no_grad():                      # keep eta_mean + h away from 0
            tr.dnn.layers[-1].bias[0] = 0.75
            tr.dnn.layers[-1].bias[3] = 0.0
    tr.train()
    losses = np.array([h[3] for h in tr.history])
    assert len(losses) >= 4 and np.all(np.isfinite(losses)), (name, losses)
    assert losses[-1] < losses[0], (name, losses)


def test_lbfgs_device_recursion_matches_torch_formulation():
    """csrc/pinn_lbfgs.hip (ring history, six launches) against lbfgs._History (torch operators): same
    direction, including after the ring has wrapped."""
    from pinn_depthestimation_amd.lbfgs import _History, _HipHistory
    P, m = 29636, 7
    g = torch.Generator().manual_seed(3)
    a, b = _History(m, torch.zeros(P, device="cuda")), _HipHistory(m, torch.zeros(P, device="cuda"))
    for it in range(2 * m + 3):
        s = (torch.randn(P, generator=g) * 1e-2).cuda()
        y = s * (0.5 + torch.rand(P, generator=g).cuda()) + 1e-3 * torch.randn(P, generator=g).cuda()
        a.push(s, y); b.push(s, y)
        grad = torch.randn(P, generator=g).cuda()
        H = float(y.dot(s) / y.dot(y))
        da, db = a.direction(grad, H), b.direction(grad, H)
        assert float((da - db).norm() / da.norm()) < 2e-5, it


@pytest.mark.parametrize("p", [0.1, 0.5])
@pytest.mark.parametrize("shape", [("ns", 3, 4, 3, 24, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
                                   ("pe", 2, 6, 8, 64, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"))])
def test_dropout_training_mode_matches_oracle_with_the_engines_mask(shape, p):
    """nn.Dropout(p > 0) after every activation in training mode (dnn.py:38, train.py:186).  The
    engine's mask is a pure function of (seed, layer, unit, point): the same mask (tests/dropout_util.py) handed to
    the oracle must give the same outputs, input derivatives, loss and parameter gradient.  Forward / jet calls run on
    the generic engine's kernels; the loss + gradient call on the generic engine AND, at padded width 64, on the fused
    tile kernel's dropout instances (k_fused<..., DROP>: the mask re-derived lane-locally in activation and adjoint)."""
    from tests.dropout_util import keep_masks
    _, d_in, d_out, L, W, gc, res, inn, outn = shape
    N, seed = 333, 20241004
    g = torch.Generator().manual_seed(17)
    params = O.init_params(O.layer_sizes(d_in, L, W, d_out), "xavier", g)
    if res == "physics_equation":
        params[-1][0] = 0.75; params[-1][3] = 0.0
    X = torch.rand(N, d_in, generator=g) * 2 - 1
    masks = [torch.from_numpy(m).double() for m in keep_masks(seed, p, L, W, N)]
    from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
    from pinn_depthestimation_amd.engine import RESIDUAL_ROLES
    desc = NetDesc(d_in, d_out, L, W, gc, dropout_p=p)
    eng = Engine(desc)
    eng.dropout_seed = seed
    flat, Xd = O.flatten(params).cuda(), X.cuda().contiguous()
    p64 = [q.double().requires_grad_(True) for q in params]
    Yo, dYo = O.jet(p64, X.double(), gc, masks=masks, p=p)
    Y, dY = eng.forward_jet(flat, Xd)
    assert (Y.cpu().double() - Yo).abs().max() < 5e-6
    assert (dY.cpu().double() - dYo).abs().max() < 5e-6 * max(1.0, float(dYo.abs().max()))
    _, out_roles, dir_roles = RESIDUAL_ROLES[res]
    lo = O.residual_loss(p64, X.double(), res, [inn.index(r) for r in dir_roles], [outn.index(r) for r in out_roles], gc,
                         masks=masks, p=p)
    go = O.flat_grad(lo, p64)
    spec = ResidualSpec.from_names(res, inn, gc, outn)
    scale = torch.full((spec.n_terms,), 1.0 / N, device="cuda")
    grad = torch.zeros(desc.n_params, device="cuda")
    sums = eng.residual_loss_grad(spec, scale, flat, Xd, grad)
    loss = float((sums.double() * scale.double()).sum())
    # tolerance: the engine's usual fp32 bars, or 4x the oracle's OWN fp32-vs-fp64 disagreement on this masked
    # network (physics_equation's 1/(rho (eta_mean + h)) is ill-conditioned, more so with half the units dropped)
    p32 = [q.clone().requires_grad_(True) for q in params]
    l32 = O.residual_loss(p32, X, res, [inn.index(r) for r in dir_roles], [outn.index(r) for r in out_roles], gc,
                          masks=[m.float() for m in masks], p=p)
    noise_l = abs(float(l32) - float(lo)) / float(lo)
    noise_g = rel_l2(O.flat_grad(l32, p32), go)
    el, eg = abs(loss - float(lo)) / float(lo), rel_l2(grad.cpu(), go)
    print(f"dropout p={p} {res}: loss err {el:.2e} (oracle fp32 {noise_l:.1e}), grad err {eg:.2e} (oracle fp32 {noise_g:.1e})")
    assert el < max(5e-6, 4 * noise_l)
    assert eg < max(3e-5, 4 * noise_g)
    # the same request on each engine that serves it: generic always; fused (tile kernel, DROP instances) at width 33..64
    from pinn_depthestimation_amd._lib import ENGINE_FUSED, ENGINE_GENERIC
    for e in (ENGINE_GENERIC, ENGINE_FUSED):
        g2 = torch.zeros(desc.n_params, device="cuda")
        if e == ENGINE_FUSED and not 32 < W <= 64:
            with pytest.raises(Exception, match="fused engine"):
                eng.residual_loss_grad(spec, scale, flat, Xd, g2, engine=e)
            continue
        s2 = eng.residual_loss_grad(spec, scale, flat, Xd, g2, engine=e)
        l2 = float((s2.double() * scale.double()).sum())
        el2, eg2 = abs(l2 - float(lo)) / float(lo), rel_l2(g2.cpu(), go)
        print(f"  engine {e}: loss err {el2:.2e}, grad err {eg2:.2e}")
        assert el2 < max(5e-6, 4 * noise_l) and eg2 < max(3e-5, 4 * noise_g)
    # another seed is another mask; p = 0 / eval is the plain network
    eng.dropout_seed = seed + 1
    assert (eng.forward(flat, Xd) - Y).abs().max() > 1e-3
    y_plain = Engine(NetDesc(d_in, d_out, L, W, gc)).forward(flat, Xd)
    assert (y_plain.cpu().double() - O.mlp_forward([q.double() for q in params], X.double())).abs().max() < 5e-6


def test_dropout_module_semantics_and_training_run():
    """dnn.DNN(p > 0): eval() is the identity network, train() draws a new mask per forward (inverted-dropout
    scaling keeps the mean), and the pinn harness trains with it (fused engines refuse dropout loudly)."""
    import dnn
    from pinn_depthestimation_amd import Engine, NetDesc, PinnError
    from pinn_depthestimation_amd.trainer import pinn
    torch.manual_seed(5)
    m = dnn.DNN([3, 64, 64, 4], 0.25, "xavier").cuda()
    x = torch.rand(4096, 3, device="cuda")
    m.eval()
    y0 = m(x)
    assert torch.equal(m(x), y0)
    m.train()
    y1, y2 = m(x), m(x)
    assert (y1 - y2).abs().max() > 1e-3                                  # two forward passes, two masks
    mean_train = torch.stack([m(x) for _ in range(64)]).mean(0)
    assert (mean_train - y0).abs().mean() < 0.1 * y0.abs().mean() + 0.05  # inverted dropout: roughly unbiased
    with pytest.raises(PinnError, match="generic engine"):      # forward-only calls with dropout: generic engine only
        Engine(NetDesc(3, 4, 2, 64, (0, 1, 2), engine=2, dropout_p=0.25)).forward(m.flat_params(), x)
    cfg = ns_config(30)
    cfg["layers"]["dropout_rate"] = 0.1
    cfg["layers"]["init_type"] = "xavier"
    z7 = load("g7_adam_ns_8x64.npz")
    tr = pinn(None, None, z7["X"][:2000], cfg, log_every=1, checkpoint_every=0)
    tr.train()
    losses = np.array([h[3] for h in tr.history])
    assert np.all(np.isfinite(losses)) and losses[-5:].mean() < losses[:5].mean()
    tr.dnn.eval()
    a = float(tr.loss_func()); b = float(tr.loss_func())
    assert a == pytest.approx(b, rel=1e-5)                               # eval mode: deterministic, no mask
