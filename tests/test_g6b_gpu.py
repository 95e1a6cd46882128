"""G6b — the training path the reference actually runs (train.py:17 imports physics_equation), against the
REFERENCE's own 200-step Adam trajectory (tests/golden/make_goldens_r3.py: dnn.DNN + weighted fidelity of
config_CMB.json:28-37 + physics.physics_equation, 12 + 243 points, Adam(1e-4) + StepLR(10000, 0.8)).

Both ways this package can run that loop are held to the same fixture:
  * the DEFAULT path: both point sets in one launch (merge_sets), Adam folded into the pass's last kernel, runs of
    iterations enqueued by pinn_adam_loop (trainer.train_adam);
  * the classic path: merge_sets = False, fold_adam = False (separate fidelity / residual launches + pinn_adam_step).

Tolerance.  The reference reproduces itself bit for bit across thread counts here (243 points: torch does not split
the work; `spread` in the fixture is 0), so the yardstick is its own fp32 rounding: the same modules run in float64
(`losses64`) differ from its fp32 run by up to 1.8e-6 (10x10) and 1.1e-3 (8x64: 1/(rho*(eta_mean+h)) amplifies fp32
rounding over the trajectory).  A different fp32 summation order (this engine) lands inside the same cloud, so every
loss must be within max(1e-5, 4 x the reference's fp32-vs-fp64 gap so far) of the reference's fp32 run.
"""
import numpy as np
import pytest
import torch

from tests.golden_util import load, state_dict

pytestmark = pytest.mark.gpu


def _cfg(hidden, width):
    return {
        "layers": {"input_features": 2, "hidden_layers": hidden, "hidden_width": width, "output_features": 6,
                   "dropout_rate": 0.0, "init_type": "xavier"},
        "adam_optimizer": {"max_it": 200, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
        "lbfgs_optimizer": {"max_it": 0, "learning_rate": 1, "max_evaluation": 6.25e4, "history_size": 100,
                            "tolerance_grad": 1e-5, "tolerance_change": 1e-7, "line_search_fn": "strong_wolfe"},
        "loss": {"weight_h_loss": 1, "weight_eta_mean_loss": 1, "weight_U_loss": 1, "weight_V_loss": 1,
                 "weight_k_loss": 1, "weight_Hrms_loss": 1, "weight_fid_loss": 1, "weight_res_loss": 1},
        "data_fidelity": {"inputs": ["x", "y"], "outputs": ["h", "U", "V", "eta_mean", "Hrms", "k"], "training_points": 12},
        "data_residual": {"inputs": {"x": {"requires_grad": ["true"]}, "y": {"requires_grad": ["true"]}},
                          "outputs": ["h", "U", "V", "eta_mean", "Hrms", "k"]},
    }


@pytest.mark.parametrize("default_path", [True, False], ids=["merged+folded", "classic"])
@pytest.mark.parametrize("net", ["10x10", "8x64"])
def test_g6b_physics_equation_adam_trajectory(net, default_path):
    from pinn_depthestimation_amd.dnn import DNN
    from pinn_depthestimation_amd.trainer import pinn
    z = load(f"g6b_adam_pe_{net}.npz")
    hidden, width = (10, 10) if net == "10x10" else (8, 64)
    model = DNN([2] + [width] * hidden + [6], 0.0, "xavier")
    model.load_state_dict(state_dict(z, "sd0/"))
    tr = pinn(z["Xf"], z["Tf"], z["Xr"], _cfg(hidden, width), dnn=model, log_every=1, checkpoint_every=0,
              fold_adam=default_path)
    tr.evaluator.merge_sets = default_path
    tr.train()
    if default_path:
        assert tr._folded_iters == 200, "the default path must be the folded one (two launches per iteration)"
    else:
        assert tr._folded_iters == 0
    got = np.array([h[1:] for h in tr.history])            # (200, 3): fidelity, residual, total
    ref, ref64 = z["losses"], z["losses64"]
    assert float(z["spread"].max()) == 0.0                 # the reference's thread-count spread (see module docstring)
    noise = np.maximum.accumulate(np.abs(ref[:, 2] - ref64[:, 2]) / np.abs(ref64[:, 2]))
    rel = np.abs(got[:, 2] - ref[:, 2]) / np.abs(ref[:, 2])
    tol = np.maximum(1e-5, 4 * noise)
    worst = int(np.argmax(rel / tol))
    print(f"G6b {net} {'default' if default_path else 'classic'}: max rel err {rel.max():.2e} (first {rel[0]:.2e}, last {rel[-1]:.2e}); "
          f"reference fp32-vs-fp64 max {noise[-1]:.2e}; tightest step {worst}: {rel[worst]:.2e} of {tol[worst]:.2e}")
    assert rel[0] < 2e-5
    assert (rel < tol).all(), (worst, rel[worst], tol[worst])
    # the two terms separately at the first and the last step
    for j, name in ((0, "fidelity"), (1, "residual")):
        for k in (0, -1):
            r = abs(got[k, j] - ref[k, j]) / abs(ref[k, j])
            assert r < max(2e-5, 8 * noise[k]), (name, k, r)
    end = torch.cat([state_dict(z, "sd_end/")[k].reshape(-1) for k in model.state_dict()])
    dw = float((model.flat_params().cpu().double() - end.double()).norm() / end.double().norm())
    assert dw < max(1e-5, 4 * noise[-1]), dw
