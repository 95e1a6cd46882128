"""The drop-in Python face (dnn.DNN, physics.*, trainer.PINN) on the GPU against vectors captured
from the reference itself (tests/golden, see make_goldens.py).  These tests are written the
way the reference's train.py uses its modules: (N,1) column tensors, torch.cat, slicing
predictions[:, i:i+1], physics_loss_calculator(...), loss.backward().

Tolerances are fp32 and stated per case; the reference's own fp32-vs-fp64 noise floor is
8.6e-8 (loss) / 1.4e-7 (grad) for Navier_Stokes and 1.5e-4 (grad) for physics_equation
(BASELINE.md §2)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

from tests.golden_util import CMB, layers_of, load, ns_config, rel_l2, state_dict  # noqa: E402

pytestmark = pytest.mark.gpu


def cols_of(X, grad_cols, device="cuda"):
    """train.py:86-88"""
    return [torch.tensor(X[:, i:i + 1].astype(np.float64), requires_grad=(i in grad_cols)).float().to(device)
            for i in range(X.shape[1])]


def flat_grad(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()


def make_model(sd):
    import dnn  # the compat shim -> pinn_depthestimation_amd.dnn
    model = dnn.DNN(layers_of(sd), 0.0, "xavier")
    model.load_state_dict(sd)
    return model.to("cuda")


def test_module_is_dropin():
    import dnn
    import physics
    from physics import physics_equation as physics_loss_calculator  # train.py:17
    m = dnn.DNN([2] + [10] * 10 + [6], 0.0, "xavier")
    assert list(m.state_dict())[:2] == ["layers.layer_0.weight", "layers.layer_0.bias"]
    assert float(m.layers.layer_3.bias.abs().sum()) == 0.0 and float(m.layers.layer_10.bias.abs().sum()) > 0
    with pytest.raises(ValueError, match="Invalid init_type"):
        dnn.DNN([2, 4, 1], 0.0, "he")
    assert callable(physics_loss_calculator) and callable(physics.compute_gradient)


@pytest.mark.parametrize("fused", [True, False])
def test_g1_g3_navier_stokes_like_train_py(fused):
    import physics
    z = load("g1_g3_ns_8x64.npz")
    model = make_model(state_dict(z))
    c = cols_of(z["X"], (0, 1, 2))
    pred = model(torch.cat(c, dim=-1))
    assert np.abs(pred.detach().cpu().numpy() - z["Y"]).max() < 2e-6
    h, zz, u, v = [pred[:, i:i + 1] for i in range(4)]
    if not fused:
        h = h * 1.0          # any arithmetic drops the JetTensor tag -> generic compute_gradient path
    for j in range(3):       # G2: the 12 compute_gradient columns
        for o, t in enumerate((h, zz, u, v)):
            g = physics.compute_gradient(t, c[j])
            assert np.abs(g.detach().cpu().numpy()[:, 0] - z["dY"][j, :, o]).max() < 5e-6
    loss = physics.Navier_Stokes(c[0], c[1], c[2], h, zz, u, v)
    model.zero_grad()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) / float(z["loss"]) < 5e-6
    assert rel_l2(flat_grad(model), z["grad"]) < 2e-5


@pytest.mark.parametrize("name,tol_g", [("g4_pe_8x64_conditioned.npz", 1e-3), ("g4_pe_10x10_rawinit.npz", 2e-1)])
def test_g4_physics_equation(name, tol_g):
    """Gradient tolerance follows the reference's own fp32-vs-fp64 disagreement (stored in the
    fixture): 1/(rho*(eta_mean+h)) makes this residual ill-conditioned."""
    from physics import physics_equation as physics_loss_calculator
    z = load(name)
    model = make_model(state_dict(z))
    c = cols_of(z["X"], (0, 1))
    pred = model(torch.cat(c, dim=-1))
    named = [pred[:, i:i + 1] for i in range(6)]
    loss = physics_loss_calculator(c[0], c[1], *named)
    model.zero_grad()
    loss.backward()
    ref_noise_l = abs(float(z["loss"]) - float(z["loss64"])) / float(z["loss64"])
    ref_noise_g = rel_l2(z["grad"], z["grad64"])
    assert abs(loss.item() - float(z["loss64"])) / float(z["loss64"]) < max(5e-6, 4 * ref_noise_l)
    assert rel_l2(flat_grad(model), z["grad64"]) < max(2e-5, 4 * ref_noise_g)
    assert ref_noise_g < tol_g
    gl = flat_grad(model)
    P = gl.numel()
    W = layers_of(state_dict(z))[-2]
    last_w = gl[P - 6 * W - 6:P - 6].reshape(6, W)
    assert torch.count_nonzero(last_w[4:6]) == 0        # Hrms, k rows: E == 0 (physics.py:106)


def test_g5_continuity_both():
    import physics
    z = load("g5_continuity_4x20.npz")
    model = make_model(state_dict(z))
    for fn in ("continuity_only", "continuity_ftemp"):
        c = cols_of(z["X"], (0, 1))
        pred = model(torch.cat(c, dim=-1))
        U, V, h = [pred[:, i:i + 1] for i in range(3)]
        loss = getattr(physics, fn)(c[0], c[1], h, U, V)
        model.zero_grad()
        loss.backward()
        assert abs(loss.item() - float(z[fn + "/loss"])) / float(z[fn + "/loss"]) < 5e-6
        assert rel_l2(flat_grad(model), z[fn + "/grad"]) < 2e-5


def test_g6_trainer_loss_func_config_cmb(tmp_path):
    from pinn_depthestimation_amd.trainer import pinn
    z = load("g6_lossfunc_cmb.npz")
    import dnn
    model = dnn.DNN(layers_of(state_dict(z)), 0.0, "xavier")
    model.load_state_dict(state_dict(z))
    tr = pinn(z["Xf"], z["Tf"], z["Xr"], CMB, dnn=model, log_dir=str(tmp_path), log_every=1)
    loss = tr.loss_func()
    fid, res, tot = (t.item() for t in tr.last)
    assert abs(fid - float(z["fid"])) / float(z["fid"]) < 5e-6
    assert abs(res - float(z["res"])) / float(z["res"]) < 2e-5
    assert abs(loss.item() - float(z["loss"])) / float(z["loss"]) < 2e-5
    assert rel_l2(tr.grad.cpu(), z["grad"]) < 1e-4
    tr.flush_log()          # log lines are written in batches (trainer.py docstring)
    lines = open(tmp_path / "log.txt").read().splitlines()
    assert lines[0] == "Epoch, Fidelity Loss, Residual Loss, Total Loss"       # train.py:167
    assert lines[1] == f"1, {fid:.5e}, {res:.5e}, {tot:.5e}"


def test_g7_adam_trajectory_200_steps():
    """BASELINE north_star: loss trajectory within 1e-5 rel of the reference CPU path."""
    from pinn_depthestimation_amd.trainer import pinn
    import dnn
    z0, z = load("g1_g3_ns_8x64.npz"), load("g7_adam_ns_8x64.npz")
    model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict(state_dict(z0))
    tr = pinn(None, None, z["X"], ns_config(200), dnn=model, log_every=1, checkpoint_every=0)
    tr.train()
    got = np.array([h[3] for h in tr.history])
    ref = z["losses"]
    rel = np.abs(got - ref) / ref
    print("adam trajectory rel err: max %.2e  first %.2e  last %.2e" % (rel.max(), rel[0], rel[-1]))
    assert rel.max() < 1e-5
    end = torch.cat([state_dict(z, "sd_end/")[k].reshape(-1) for k in model.state_dict()])
    assert rel_l2(model.flat_params().cpu(), end) < 1e-5


def test_g8_lbfgs_from_adam_end_state():
    from pinn_depthestimation_amd.trainer import pinn
    import dnn
    z7, z = load("g7_adam_ns_8x64.npz"), load("g8_lbfgs_ns_8x64.npz")
    model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict(state_dict(z7, "sd_end/"))
    tr = pinn(None, None, z7["X"][:2000], ns_config(0, lbfgs_it=50), dnn=model, log_every=1, checkpoint_every=0)
    tr.train()
    got = np.array([h[3] for h in tr.history])
    ref = z["losses"]
    assert abs(got[0] - ref[0]) / ref[0] < 5e-6
    # Line-search decisions amplify rounding, and the reference does it to itself: G8s is the same
    # torch.optim.LBFGS run at 1, 2 and 4 threads instead of 8 — its closure losses differ from G8's by up to
    # 6.4e-5 over the first 10 evaluations, 9.7e-5 over the first 20 and 1.2e-2 by the end.  Asserted: 4x the
    # reference's own spread on the first 10 (2.6e-4; SURVEY §8c's 1e-4 is below what the reference does to
    # itself at 4x) and on the first 20, and the end of the descent within 4x its end-of-run spread.
    spread = load("g8s_lbfgs_thread_spread.npz")["spread"]
    n = min(len(got), len(ref))
    rel = np.abs(got[:n] - ref[:n]) / ref[:n]
    print("G8 torch-LBFGS closure losses vs reference: first 10 %.2e, first 20 %.2e, all %.2e (reference's own "
          "thread spread: %.2e / %.2e / %.2e)" % (rel[:10].max(), rel[:20].max(), rel.max(), np.nanmax(spread[:10]),
                                                    np.nanmax(spread[:20]), np.nanmax(spread)))
    assert n >= 20
    assert rel[:10].max() < max(1e-4, 4 * np.nanmax(spread[:10]))
    assert rel[:20].max() < 4 * np.nanmax(spread[:20])
    assert abs(np.log(got[-1] / ref[-1])) < max(0.05, 4 * np.nanmax(spread))


def test_g8b_scipy_lbfgsb_trajectory():
    """a9 (second half): the SciPy L-BFGS-B stage over the flat closure, against the same
    scipy.optimize.minimize run over a closure built from the REFERENCE's dnn.DNN / physics.Navier_Stokes
    (make_goldens_r2.py g8b; start = G7 end state, N = 2000, maxcor 50, maxls 50, 50 iterations).
    Every closure evaluation is compared.  SURVEY §8c asked for 1e-4 on the leading iterates and 5 % at the end;
    the reference cannot meet that against ITSELF: re-run at 1, 2 and 4 threads (g8s: only the summation order of
    its fp32 kernels changes) its evaluated losses move by up to 6.0e-4 within the first 12 evaluations, 4.6e-1
    later, and its final loss by 27 % — L-BFGS-B's line search amplifies 1e-7 differences by four orders of
    magnitude on this problem.  Asserted: 1e-4 on the first 5 evaluations (before the amplification sets in),
    20x the reference's own spread on the first 12, the same iteration count, and a final loss within a factor 2."""
    from pinn_depthestimation_amd.lbfgsb import LBFGSBOptimizer
    from pinn_depthestimation_amd.trainer import pinn
    import dnn
    z7, z = load("g7_adam_ns_8x64.npz"), load("g8b_scipy_lbfgsb_ns_8x64.npz")
    model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict(state_dict(z7, "sd_end/"))
    tr = pinn(None, None, z7["X"][:2000], ns_config(0), dnn=model, log_every=1, checkpoint_every=0)
    opt = LBFGSBOptimizer(tr, {"maxiter": 50, "maxfun": 50000})
    res = opt.minimize()
    got, ref = np.array(opt.losses[:-1]), z["evals"]            # (the last entry re-evaluates the end point)
    n = min(len(got), len(ref))
    rel = np.abs(got[:n] - ref[:n]) / ref[:n]
    print("G8b SciPy L-BFGS-B: %d / %d evaluations, nit %d / %d; rel diff first 12 evals %.2e, all %.2e; final %.4e vs %.4e"
          % (len(got), len(ref), res.nit, int(z["nit"]), rel[:12].max(), rel.max(), res.fun, float(z["fun"])))
    zs = load("g8s_lbfgs_thread_spread.npz")
    print("   reference's own thread spread: first 12 %.2e, all %.2e, final loss %.2e" %
          (np.nanmax(zs["scipy_spread"][:12]), np.nanmax(zs["scipy_spread"]), float(zs["scipy_end_spread"])))
    assert abs(got[0] - ref[0]) / ref[0] < 5e-6
    assert n >= 12 and rel[:5].max() < 1e-4
    assert rel[:12].max() < 20 * np.nanmax(zs["scipy_spread"][:12])
    assert res.nit == int(z["nit"])
    assert abs(np.log(res.fun / float(z["fun"]))) < np.log(2.0)
    assert res.fun < 0.05 * got[0]                               # and it is a descent: two orders below the start


def test_g9_newmethod_on_data_at50k_columns():
    from pinn_depthestimation_amd.trainer import pinn
    import dnn
    z, zx = load("g9_newmethod_at50k.npz"), load("g9x_newmethod_fp64.npz")
    T = np.concatenate([z["U"], z["V"]], 1)
    for tag, hidden, width, tol in (("8x64", 8, 64, 1e-5), ("100x20", 100, 20, 1e-4)):
        cfg = {"layers": {"input_features": 2, "hidden_layers": hidden, "hidden_width": width, "output_features": 3,
                          "dropout_rate": 0.0, "init_type": "xavier"},
               "adam_optimizer": {"max_it": 5, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
               "lbfgs_optimizer": {"max_it": 0},
               "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
               "data": {"inputs": {"x": {"requires_grad": ["true"]}, "y": {"requires_grad": ["true"]}},
                        "trues": ["U", "V"], "unknowns": ["h"]}}
        sd = state_dict(z, f"{tag}/sd/")
        model = dnn.DNN(layers_of(sd), 0.0, "xavier")
        model.load_state_dict(sd)
        tr = pinn(z["X"], T, z["X"], cfg, dnn=model, log_every=1, checkpoint_every=0)
        tr.loss_func()
        # first iteration against the reference's fp64 evaluation of the same state (g9x), tolerance = the
        # engine's usual fp32 bars or 4x the reference's own fp32-vs-fp64 disagreement, whichever is larger
        e_fid = abs(tr.last[0].item() - float(zx[f"{tag}/fid64"])) / float(zx[f"{tag}/fid64"])
        e_res = abs(tr.last[1].item() - float(zx[f"{tag}/res64"])) / float(zx[f"{tag}/res64"])
        e_grad = rel_l2(tr.grad.cpu(), zx[f"{tag}/grad64"])
        print(f"G9 {tag}: vs reference fp64: fid {e_fid:.2e} res {e_res:.2e} grad {e_grad:.2e} (reference fp32: "
              f"{float(zx[f'{tag}/ref_fid_err']):.1e} {float(zx[f'{tag}/ref_res_err']):.1e} {float(zx[f'{tag}/ref_grad_err']):.1e})")
        assert e_fid < max(5e-6, 4 * float(zx[f"{tag}/ref_fid_err"]))
        assert e_res < max(5e-6, 4 * float(zx[f"{tag}/ref_res_err"]))
        assert e_grad < max(2e-5, 4 * float(zx[f"{tag}/ref_grad_err"]))
        tr.iter, tr.history = 0, []
        tr.train()
        got = np.array([h[3] for h in tr.history])
        assert np.max(np.abs(got - z[f"{tag}/losses"]) / z[f"{tag}/losses"]) < tol


def test_cpu_tensor_is_refused_loudly():
    import dnn
    from pinn_depthestimation_amd import PinnError
    m = dnn.DNN([2, 8, 8, 1], 0.0, "xavier")
    with pytest.raises(PinnError, match="no CPU path"):
        m(torch.zeros(4, 2))


def test_g7b_adam_trajectory_1000_steps():
    """Longer horizon than G7: 1000 Adam steps (StepLR 250 / 0.8), N = 4096, against the reference's own
    CPU run.  Observed max relative difference 3.5e-6 over the 1000 steps; asserted 1e-5
    (BASELINE north_star: "loss within 1e-5 rel of reference")."""
    from pinn_depthestimation_amd.trainer import pinn
    import dnn
    z0, z = load("g1_g3_ns_8x64.npz"), load("g7b_adam_1000_ns_8x64.npz")
    model = dnn.DNN([3] + [64] * 8 + [4], 0.0, "xavier")
    model.load_state_dict(state_dict(z0))
    tr = pinn(None, None, z["X"], ns_config(1000, step=250), dnn=model, log_every=1, checkpoint_every=0)
    tr.train()
    got = np.array([h[3] for h in tr.history])
    ref = z["losses"]
    rel = np.abs(got - ref) / ref
    print("1000-step trajectory rel err: max %.2e | @200 %.2e | @500 %.2e | @1000 %.2e" % (rel.max(), rel[:200].max(), rel[:500].max(), rel[-1]))
    assert rel.max() < 1e-5
