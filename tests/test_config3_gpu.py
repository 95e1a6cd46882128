"""BASELINE.json configs[3] AS CONFIGURED: 3 -> 12 x 256 tanh -> 4, Navier_Stokes, bf16 MFMA operands with fp32
accumulate / residual (PINN_PREC_BF16 on the wide engine), against vectors the REFERENCE produced
(tests/golden/make_goldens_r2.py: G10 = dnn.DNN + physics.Navier_Stokes in fp32 and fp64 at N = 2000, G10b = 100
Adam steps of the reference in fp32).  Weights come from tests/golden/synth.py (seed in the fixture).

Tolerances:
  fp32 mode   loss 3e-6, gradient 3e-5 rel-L2 vs the reference's fp64 run (its own fp32-vs-fp64 noise at this
              shape is 8.7e-8 / 1.6e-7, stored in the fixture); forward 3e-6 abs vs the reference's fp32 Y.
  bf16 mode   this is a TOLERANCE, not fp32 parity: the jets between layers carry 8 significant bits (the weights are
              split hi + lo and multiply exactly).  Measured against the reference's fp64 run at N = 2000:
              2.8e-3 (loss) / 3.3e-3 (gradient), asserted at 5e-3 / 5e-3 (1.5x to 1.8x of what is measured; the margin
              covers box-to-box and launch-geometry differences of the summation order, which move the third digit).
              At 2^21 points, bf16 mode against fp32 mode of the same engine: 3.1e-3 / 3.5e-3, asserted at 5.5e-3
              (1.55x).  north_star's "loss within 1e-5 rel" is met by the fp32 modes only (G7 / G7b / G10b: 7.8e-6);
              bf16 mode's trajectory check (G10b) is a descent check — every loss within 25 % (measured 15 %).
"""
import numpy as np
import pytest
import torch

from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd._lib import ENGINE_WIDE, PREC_BF16, PREC_F32
from tests.golden import synth
from tests.golden_util import load, rel_l2

pytestmark = pytest.mark.gpu

LAYERS = [3] + [256] * 12 + [4]
DESC = NetDesc(3, 4, 12, 256, (0, 1, 2), engine=ENGINE_WIDE)
SPEC = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), (0, 1, 2), ("h", "z", "u", "v"))


def g10_flat(seed):
    return torch.from_numpy(np.concatenate([a.reshape(-1) for a in synth.xavier_params(LAYERS, int(seed))])).cuda()


def loss_and_grad(eng, flat, X, n_global=None):
    N = X.shape[0]
    scale = torch.full((3,), 1.0 / (n_global or N), device="cuda")
    grad = torch.zeros(DESC.n_params, device="cuda")
    sums = eng.residual_loss_grad(SPEC, scale, flat, X, grad)
    return float((sums.double() * scale.double()).sum()), grad, sums


def test_g10_fp32_wide_matches_reference():
    z = load("g10_ns_12x256.npz")
    flat, X = g10_flat(z["seed"]), torch.from_numpy(z["X"]).cuda()
    eng = Engine(DESC.with_(precision=PREC_F32))
    Y = eng.forward(flat, X)
    assert np.abs(Y.cpu().numpy() - z["Y32"]).max() < 3e-6
    loss, grad, _ = loss_and_grad(eng, flat, X)
    el = abs(loss - float(z["loss64"])) / float(z["loss64"])
    eg = rel_l2(grad.cpu(), z["grad64"])
    print(f"G10 fp32 wide: loss rel err {el:.2e} (reference fp32: {float(z['ref_loss_err']):.1e}), "
          f"grad rel-L2 {eg:.2e} (reference fp32: {float(z['ref_grad_err']):.1e})")
    assert el < 3e-6 and eg < 3e-5


def test_g10_bf16_mode_matches_reference_within_bf16_tolerance():
    z = load("g10_ns_12x256.npz")
    flat, X = g10_flat(z["seed"]), torch.from_numpy(z["X"]).cuda()
    eng = Engine(DESC.with_(precision=PREC_BF16))
    loss, grad, _ = loss_and_grad(eng, flat, X)
    el = abs(loss - float(z["loss64"])) / float(z["loss64"])
    eg = rel_l2(grad.cpu(), z["grad64"])
    print(f"G10 bf16 mode: loss rel err {el:.2e}, grad rel-L2 {eg:.2e}")
    assert el < 5e-3 and eg < 5e-3
    assert eg > 1e-5                      # it really is the reduced-precision path
    # the forward of the network (dnn.py:54-55) in bf16 mode: outputs are O(1), bf16 operand rounding 2^-9
    Y = eng.forward(flat, X)
    assert np.abs(Y.cpu().numpy() - z["Y32"]).max() < 2e-2


def _adam_trajectory(precision, X, flat0, steps):
    eng = Engine(DESC.with_(precision=precision))
    flat = flat0.clone()
    P, N = DESC.n_params, X.shape[0]
    scale = torch.full((3,), 1.0 / N, device="cuda")
    grad, m, v = (torch.zeros(P, device="cuda") for _ in range(3))
    sums = torch.zeros(steps, 3, device="cuda")
    for i in range(steps):                                   # train.py:189-193
        grad.zero_()
        eng.residual_loss_grad(SPEC, scale, flat, X, grad, sums=sums[i])
        eng.adam_step(flat, grad, m, v, i + 1, 1e-4)
    return (sums.double().sum(1) / N).cpu().numpy()


def test_g10b_adam_trajectory_fp32_and_bf16_drift():
    """100 Adam(1e-4) steps from the G10 state, against the REFERENCE's fp32 CPU trajectory.  At this size Adam's
    first steps overshoot (loss 0.040 -> 0.52 at step 10, back to 2e-4 by step 90), so the trajectory is a
    sensitive probe.  fp32 wide: within 1e-4 of the reference over all 100 steps (1e-5 over the first 10; the
    chaotic overshoot amplifies rounding after that).  bf16 mode: same descent — every loss within 25 % of the
    reference's, the first step within 5e-3 — measured drift is printed."""
    z, zb = load("g10_ns_12x256.npz"), load("g10b_adam_ns_12x256.npz")
    flat0, X = g10_flat(z["seed"]), torch.from_numpy(z["X"]).cuda()
    ref = zb["losses"]
    got32 = _adam_trajectory(PREC_F32, X, flat0, len(ref))
    r32 = np.abs(got32 - ref) / ref
    print("fp32 wide vs reference: max rel %.2e (first 10: %.2e)" % (r32.max(), r32[:10].max()))
    got16 = _adam_trajectory(PREC_BF16, X, flat0, len(ref))
    r16 = np.abs(got16 - ref) / ref
    print("bf16 mode vs reference: first %.2e, max over 100 steps %.2e, last %.2e" % (r16[0], r16.max(), r16[-1]))
    assert r32[:10].max() < 1e-5 and r32.max() < 1e-4
    assert r16[0] < 5e-3 and r16.max() < 0.25
    assert got16[-1] < 0.05 * got16[10]                       # bf16 mode trains: same two orders of descent


def test_config3_full_size_properties():
    """configs[3]'s per-GPU size, 2 097 152 points (the oracle cannot run it): size-independent properties.
    (a) additivity: sums over a split of the points add up to the sums over the whole set, and so does the
        gradient (both modes; atomics reorder the additions: 2e-5);
    (b) bf16 mode agrees with fp32 mode of the same engine within the bf16 tolerance of the G10 test."""
    N = 2_097_152
    g = torch.Generator().manual_seed(33)
    flat = g10_flat(1010)
    X = (torch.rand(N, 3, generator=g) * 2 - 1).cuda()
    cut = 777_777                                             # not a multiple of 16: ragged tiles on both sides
    res = {}
    for prec in (PREC_F32, PREC_BF16):
        eng = Engine(DESC.with_(precision=prec))
        loss, grad, sums = loss_and_grad(eng, flat, X)
        _, ga, sa = loss_and_grad(eng, flat, X[:cut].contiguous(), N)
        _, gb, sb = loss_and_grad(eng, flat, X[cut:].contiguous(), N)
        assert torch.allclose(sa + sb, sums, rtol=2e-5), (prec, sa + sb, sums)
        assert rel_l2((ga + gb).cpu(), grad.cpu()) < 2e-5
        assert bool(torch.isfinite(grad).all())
        res[prec] = (loss, grad.cpu())
    el = abs(res[PREC_BF16][0] - res[PREC_F32][0]) / res[PREC_F32][0]
    eg = rel_l2(res[PREC_BF16][1], res[PREC_F32][1])
    print(f"2^21 points: bf16 vs fp32 mode: loss {el:.2e}, grad {eg:.2e}")
    assert el < 5.5e-3 and eg < 5.5e-3       # measured 3.1e-3 / 3.5e-3 (module docstring)
