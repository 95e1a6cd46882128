"""Seeded random sweep over what the fused engine accepts — network shape (depth 1..40, width 3..64, extra input / output
columns, shuffled column order, an extra differentiated input), residual, request kind (residual / one pass with fidelity
columns / split point set) and point count (1 .. 150 001: one tile, ragged tails, the cooperative range, one-tile
batches, full batches) — every fused kernel that takes the case (AUTO's pick, tile, cooperative, batch) against the
GENERIC engine (one thread per point, no MFMA, no shared code beyond residuals.h; itself pinned by the oracle in
test_engine_gpu.py) on the same device buffers.  fp32 both sides: sums 1e-4 (summation order), gradient 1e-4 rel-L2 — the
generic kernels' own fp32 accumulation is the noisier side (continuity_only, 20 000 points: generic 2.2e-5 from the fp64
oracle, the fused kernels 7e-8; tests/devtools/sweep_debug.py) and a wrong kernel is off by >= 1e-2.

The committed seed ranges run in seconds; the same draws over 1 200 seeds (ranges widened by hand) pass as well.
physics_equation cases scale the output layer by 0.25: with eta_mean + h near 0 the residual's 1 / (rho (eta_mean + h))
term makes two fp32 evaluations differ by 1e-2 (the reference itself: G4's raw-init fixture, 2e-1), which says nothing
about the kernels.

Found by this sweep (round 3): a residual spec's UNUSED role entries (zero-filled) claimed output column 0 in the
engines' scatter tables whenever no real role sat in column 0 — garbage output adjoints for networks whose first
output column is not one of the residual's roles.  pinn_abi.hip now hands the engines a normalised spec."""
import random

import pytest
import torch

from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd._lib import (ENGINE_AUTO, ENGINE_FUSED_BATCH, ENGINE_FUSED_COOP, ENGINE_FUSED_TILE,
                                           ENGINE_GENERIC, ENGINE_WIDE)
from pinn_depthestimation_amd.dnn import init_flat_params
from pinn_depthestimation_amd.engine import RESIDUAL_ROLES, PinnError

pytestmark = pytest.mark.gpu

WIDTHS = [3, 7, 10, 12, 16, 17, 20, 24, 28, 32, 33, 40, 48, 57, 64]
POINTS = [1, 15, 37, 243, 700, 4097, 9600, 20000, 70001, 150001]
KERNELS = {"auto": ENGINE_AUTO, "tile": ENGINE_FUSED_TILE, "coop": ENGINE_FUSED_COOP, "batch": ENGINE_FUSED_BATCH}


WIDE_WIDTHS = [65, 72, 100, 128, 129, 200, 256]
WIDE_KERNELS = {"auto": ENGINE_AUTO, "wide": ENGINE_WIDE}


def draw(seed, wide=False):
    r = random.Random(seed)
    res = r.choice(sorted(RESIDUAL_ROLES))
    _, out_roles, dir_roles = RESIDUAL_ROLES[res]
    inn = list(dir_roles) + [f"in{i}" for i in range(r.choice([0, 0, 1, 2]))]
    outn = list(out_roles) + [f"out{i}" for i in range(r.choice([0, 0, 1, 3]))]
    r.shuffle(inn); r.shuffle(outn)
    gc = sorted(inn.index(d) for d in dir_roles)
    extra = [i for i in range(len(inn)) if i not in gc]
    if extra and len(gc) < 3 and r.random() < 0.4:
        gc = sorted(gc + [r.choice(extra)])          # a differentiated input no residual role uses (k grows by one)
    L = r.choice([1, 2, 3, 4, 6, 8, 10, 12, 14, 40]) if seed % 7 else r.choice([1, 2, 3])
    W = r.choice(WIDTHS)
    N = r.choice(POINTS)
    if L == 40: W, N = min(W, 24), min(N, 20000)
    if wide:
        L, W, N = r.choice([1, 2, 3, 5, 12]), r.choice(WIDE_WIDTHS), r.choice(POINTS[:7])
    kind = r.choice(["residual", "residual", "onepass", "split"])
    return res, inn, outn, tuple(gc), L, W, N, kind


@pytest.mark.parametrize("seed", range(56))
def test_random_case_against_the_generic_engine(seed):
    check(seed, draw(seed), KERNELS)


@pytest.mark.parametrize("seed", range(1000, 1024))
def test_random_wide_case_against_the_generic_engine(seed):
    """The same sweep over 64 < width <= 256 (fp32 mode): the wide engine's three kernels per layer."""
    check(seed, draw(seed, wide=True), WIDE_KERNELS)


def check(seed, case, kernels):
    res, inn, outn, gc, L, W, N, kind = case
    d_in, d_out = len(inn), len(outn)
    g = torch.Generator().manual_seed(100 + seed)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    if res == "continuity_only":
        X[:, inn.index("x")] *= 40                   # x < 25.5 selects a real subset
    base = NetDesc(d_in, d_out, L, W, gc)
    params = init_flat_params(base.layers, "xavier", g).cuda()
    nP = base.n_params
    params[nP - d_out:] = torch.rand(d_out, generator=g).cuda() * 0.2
    if res == "physics_equation":
        params[nP - d_out + outn.index("h")] = 0.75
        params[nP - d_out + outn.index("k")] = 0.5
        params[nP - d_out - W * d_out:nP - d_out] *= 0.25     # keep eta_mean + h away from 0: 1 / (rho (eta_mean + h)) is singular there
    n_fid = {"residual": 0, "onepass": N, "split": max(1, N // 5)}[kind]
    fid_cols = sorted(random.Random(seed).sample(range(d_out), min(d_out, 1 + seed % 3)))
    T = torch.rand(n_fid, len(fid_cols), generator=g).cuda()
    n_res = N - n_fid if kind == "split" else N
    if kind == "split" and n_res == 0:
        kind, n_res, n_fid = "residual", N, 0
    desc_of = lambda e: NetDesc(d_in, d_out, L, W, gc, engine=e)
    spec = ResidualSpec.from_names(res, inn, gc, outn)
    scale = torch.full((spec.n_terms,), 1.0 / max(n_res, 1), device="cuda") * torch.tensor([1.0, 0.5, 2.0, 1.5][:spec.n_terms], device="cuda")
    cscale = torch.full((len(fid_cols),), 0.7 / max(n_fid, 1), device="cuda")

    def run(e):
        eng, grad = Engine(desc_of(e)), torch.zeros(nP, device="cuda")
        if kind == "residual":
            s, c = eng.residual_loss_grad(spec, scale, params, X, grad), torch.zeros(0, device="cuda")
        elif kind == "onepass":
            s, c = eng.residual_mse_loss_grad(spec, scale, T, fid_cols, cscale, params, X, grad)
        else:
            s, c = eng.residual_mse_split_loss_grad(spec, scale, T, fid_cols, cscale, params, X, n_res, grad)
        torch.cuda.synchronize()
        return torch.cat([s, c]).double().cpu(), grad.double().cpu()

    s0, g0 = run(ENGINE_GENERIC)
    assert torch.isfinite(s0).all() and torch.isfinite(g0).all() and g0.norm() > 0
    ran, worst = [], 0.0
    for tag, e in kernels.items():
        try:
            s1, g1 = run(e)
        except PinnError as err:
            assert tag != "auto", err                 # AUTO takes everything (falls back to the generic kernels itself)
            assert "support" in str(err) or "engine" in str(err), err
            continue
        ran.append(tag)
        rel = float((g1 - g0).norm() / g0.norm())
        worst = max(worst, rel)
        assert torch.allclose(s1, s0, rtol=1e-4, atol=1e-9), (tag, s1, s0)
        assert rel < 1e-4, (tag, rel)
    print(f"seed {seed}: {res} {d_in}->{L}x{W}->{d_out} k={len(gc)} N={N} {kind}: {'/'.join(ran)} (worst gradient rel-L2 {worst:.1e})")
    assert ("tile" in ran) if W <= 64 else ("wide" in ran)


def draw_net(seed):
    """A random network and point set with no residual attached: any input / output count, 0..3 differentiated inputs."""
    r = random.Random(seed)
    d_in, d_out = r.choice([1, 2, 3, 4, 6]), r.choice([1, 2, 3, 4, 5, 6, 9])
    k = r.choice([0, 0, 1, 2, 3, 3])
    gc = tuple(sorted(r.sample(range(d_in), min(k, d_in))))
    W = r.choice(WIDTHS + WIDE_WIDTHS[:5] if seed % 3 else WIDTHS)
    L = r.choice([1, 2, 3, 5, 8, 12])
    N = r.choice(POINTS[:8])
    if seed % 4 == 0 and W <= 64:
        N = r.choice([150001, 300001])      # the plain forward's four-tiles-per-wave kernel (pinn_fused_plain.hip) takes over up here
    return d_in, d_out, gc, L, W, N


ALL_KERNELS = dict(KERNELS, wide=ENGINE_WIDE)


@pytest.mark.parametrize("seed", range(2000, 2048))
def test_random_forward_and_jet_against_the_generic_engine(seed):
    """pinn_forward / pinn_forward_jet (DNN.forward and the compute_gradient columns, dnn.py:54-55, physics.py:6-15): Y
    and dY of every engine that takes the network against the generic kernels, 5e-6 of the largest entry."""
    d_in, d_out, gc, L, W, N = draw_net(seed)
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    base = NetDesc(d_in, d_out, L, W, gc)
    params = init_flat_params(base.layers, "xavier", g).cuda()
    ref = Engine(base.with_(engine=ENGINE_GENERIC))
    Y0 = ref.forward(params, X)
    Yj0, dY0 = ref.forward_jet(params, X) if gc else (Y0, None)
    assert torch.isfinite(Y0).all() and torch.equal(Y0, Yj0)
    ran = []
    for tag, e in ALL_KERNELS.items():
        eng = Engine(base.with_(engine=e))
        try:
            Y1 = eng.forward(params, X)
            Yj1, dY1 = eng.forward_jet(params, X) if gc else (Y1, None)
        except PinnError as err:
            assert tag != "auto", err
            continue
        torch.cuda.synchronize()
        ran.append(tag)
        tol = 5e-6 * max(1.0, float(Y0.abs().max()))
        assert float((Y1 - Y0).abs().max()) < tol and float((Yj1 - Y0).abs().max()) < tol, tag
        if gc:
            assert float((dY1 - dY0).abs().max()) < 5e-6 * max(1.0, float(dY0.abs().max())), tag
    print(f"seed {seed}: {d_in}->{L}x{W}->{d_out} k={len(gc)} N={N}: {'/'.join(ran)}")
    assert len(ran) >= 1


@pytest.mark.parametrize("seed", range(3000, 3032))
def test_random_fidelity_only_request_against_the_generic_engine(seed):
    """pinn_mse_loss_grad (train.py:136-141: the fidelity term alone, k = 0 networks included)."""
    d_in, d_out, gc, L, W, N = draw_net(seed)
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    base = NetDesc(d_in, d_out, L, W, gc)
    params = init_flat_params(base.layers, "xavier", g).cuda()
    cols = sorted(random.Random(seed).sample(range(d_out), random.Random(seed + 1).randint(1, min(d_out, 6))))
    T = torch.rand(N, len(cols), generator=g).cuda()
    cscale = (torch.rand(len(cols), generator=g) + 0.5).cuda() / N
    out = {}
    for tag, e in dict(ALL_KERNELS, generic=ENGINE_GENERIC).items():
        grad = torch.zeros(base.n_params, device="cuda")
        try:
            s = Engine(base.with_(engine=e)).mse_loss_grad(params, X, T, cols, cscale, grad)
        except PinnError as err:
            assert tag not in ("auto", "generic"), err
            continue
        torch.cuda.synchronize()
        out[tag] = (s.double().cpu(), grad.double().cpu())
    s0, g0 = out.pop("generic")
    assert g0.norm() > 0
    for tag, (s1, g1) in out.items():
        assert torch.allclose(s1, s0, rtol=1e-4), (tag, s1, s0)
        assert float((g1 - g0).norm() / g0.norm()) < 1e-4, tag
    print(f"seed {seed}: {d_in}->{L}x{W}->{d_out} k={len(gc)} N={N} cols {cols}: {'/'.join(out)}")


@pytest.mark.parametrize("seed", range(4000, 4024))
def test_random_folded_adam_iterations_against_the_separate_calls(seed):
    """pinn_loss_grad_adam_step on random fused-engine cases (AUTO's kernel pick: cooperative / tile / batch, every
    packing): four folded iterations against loss call + pinn_adam_step, with a torch write in between."""
    res, inn, outn, gc, L, W, N, kind = draw(seed)
    N = max(N, 16)
    d_in, d_out = len(inn), len(outn)
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    if res == "continuity_only":
        X[:, inn.index("x")] *= 40
    desc = NetDesc(d_in, d_out, L, W, gc)
    flat0 = init_flat_params(desc.layers, "xavier", g).cuda()
    nP = desc.n_params
    flat0[nP - d_out:] = torch.rand(d_out, generator=g).cuda() * 0.2
    if res == "physics_equation":
        flat0[nP - d_out + outn.index("h")] = 0.75
        flat0[nP - d_out + outn.index("k")] = 0.5
        flat0[nP - d_out - W * d_out:nP - d_out] *= 0.25     # keep eta_mean + h away from 0: 1 / (rho (eta_mean + h)) is singular there
    n_fid = {"residual": 0, "onepass": N, "split": max(1, N // 5)}[kind]
    fid = sorted(random.Random(seed).sample(range(d_out), min(d_out, 1 + seed % 3))) if kind != "residual" else []
    T = torch.rand(n_fid, len(fid), generator=g).cuda() if fid else None
    n_res = N - n_fid if kind == "split" else N
    spec = ResidualSpec.from_names(res, inn, gc, outn)
    scale = torch.full((spec.n_terms,), 1.0 / n_res, device="cuda")
    cscale = torch.full((len(fid),), 0.7 / max(n_fid, 1), device="cuda") if fid else None
    runs = {}
    for mode in ("classic", "folded"):
        eng = Engine(desc)
        th, m, v, grad = flat0.clone(), torch.zeros(nP, device="cuda"), torch.zeros(nP, device="cuda"), torch.zeros(nP, device="cuda")
        ts, cs = torch.zeros(spec.n_terms, device="cuda"), torch.zeros(max(len(fid), 1), device="cuda")[:len(fid)]
        g1 = None
        for step in range(1, 5):
            if step == 3:
                th.mul_(1.0 + 1e-3)
            if mode == "folded":
                ok = eng.loss_grad_adam_step(spec, scale, th, X, n_res if kind == "split" else -1 if fid else N, grad, m, v, step, 1e-3,
                                             T=T, out_col=fid, col_scale=cscale, term_sums=ts, col_sums=cs if fid else None)
                if not ok:
                    pytest.skip("not a one-pass request of the fused engine")
            else:
                grad.zero_()
                if not fid:
                    eng.residual_loss_grad(spec, scale, th, X, grad, sums=ts)
                elif kind == "onepass":
                    eng.residual_mse_loss_grad(spec, scale, T, fid, cscale, th, X, grad, term_sums=ts, col_sums=cs)
                else:
                    eng.residual_mse_split_loss_grad(spec, scale, T, fid, cscale, th, X, n_res, grad, term_sums=ts, col_sums=cs)
                eng.adam_step(th, grad, m, v, step, 1e-3)
            if step == 1:
                g1 = (grad.clone(), ts.clone(), cs.clone())
        torch.cuda.synchronize()
        runs[mode] = (th, m, v, g1)
    a, b = runs["classic"], runs["folded"]
    assert float((a[3][0] - b[3][0]).norm() / a[3][0].norm()) < 5e-6
    assert torch.allclose(a[3][1], b[3][1], rtol=1e-5) and torch.allclose(a[3][2], b[3][2], rtol=1e-5)
    # Adam's first steps move every parameter by ~lr whatever the gradient's size: entries whose gradient is at rounding
    # level may step the other way, so compare in units of the step (4 iterations x 1e-3)
    assert float((a[0] - b[0]).abs().max()) < 4.1e-3 and float((a[0] - b[0]).norm() / (a[0] - flat0).norm()) < 2e-2
    assert bool(torch.isfinite(b[0]).all())
    print(f"seed {seed}: {res} {d_in}->{L}x{W}->{d_out} k={len(gc)} N={N} {kind}")


@pytest.mark.parametrize("seed", range(5000, 5024))
def test_random_bf16_mode_case_against_fp32_mode(seed):
    """precision = bf16 (the chain engine, configs[3]'s mode) over random wide shapes: every path the shape selects —
    first layer folded into the forward chain or not (d_in <= 3), the streaming output-layer kernels or the wide
    engine's (d_out <= 4), k = 2 / 3, ragged tile counts — against fp32 mode on the same buffers.  bf16 mode is a
    TOLERANCE mode (jets carry 8 significant bits): loss sums 3e-2, gradient 1e-1 rel-L2 here (shallow random nets sit at
    1e-3..1e-2); what this sweep is after is a path that reads or writes the wrong thing, which is off by O(1) or NaN."""
    r = random.Random(seed)
    res, inn, outn, gc, _, _, _, kind = draw(seed)
    L, W, N = r.choice([2, 3, 5, 8, 12]), r.choice(WIDE_WIDTHS), r.choice([15, 243, 700, 4097, 9600, 20000])
    d_in, d_out = len(inn), len(outn)
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    if res == "continuity_only":
        X[:, inn.index("x")] *= 40
    base = NetDesc(d_in, d_out, L, W, gc)
    params = init_flat_params(base.layers, "xavier", g).cuda()
    nP = base.n_params
    params[nP - d_out:] = torch.rand(d_out, generator=g).cuda() * 0.2
    if res == "physics_equation":
        params[nP - d_out + outn.index("h")] = 0.75
        params[nP - d_out + outn.index("k")] = 0.5
        params[nP - d_out - W * d_out:nP - d_out] *= 0.25     # keep eta_mean + h away from 0: 1 / (rho (eta_mean + h)) is singular there
    spec = ResidualSpec.from_names(res, inn, gc, outn)
    scale = torch.full((spec.n_terms,), 1.0 / N, device="cuda")
    out = {}
    for tag, prec in (("fp32", 0), ("bf16", 1)):
        grad = torch.zeros(nP, device="cuda")
        eng = Engine(base.with_(precision=prec))
        s = eng.residual_loss_grad(spec, scale, params, X, grad)
        Y, dY = eng.forward_jet(params, X)
        torch.cuda.synchronize()
        out[tag] = (s.double().cpu(), grad.double().cpu(), Y.double().cpu(), dY.double().cpu())
    (s0, g0, Y0, dY0), (s1, g1, Y1, dY1) = out["fp32"], out["bf16"]
    rel = float((g1 - g0).norm() / g0.norm())
    ey, edy = float((Y1 - Y0).abs().max()), float((dY1 - dY0).abs().max() / dY0.abs().max().clamp_min(1e-30))
    print(f"seed {seed}: {res} {d_in}->{L}x{W}->{d_out} k={len(gc)} N={N}: sums {float(((s1 - s0) / s0.clamp_min(1e-30)).abs().max()):.1e} "
          f"gradient {rel:.1e} Y {ey:.1e} dY {edy:.1e}")
    assert torch.isfinite(g1).all() and torch.isfinite(s1).all()
    live = s0.abs() > 1e-12
    assert torch.allclose(s1[live], s0[live], rtol=3e-2), (s1, s0)
    assert rel < 1e-1
    assert ey < 2e-2 * max(1.0, float(Y0.abs().max())) and edy < 5e-2


@pytest.mark.parametrize("seed", range(6000, 6016))
def test_random_dropout_case_fused_instance_against_the_generic_engine(seed):
    """nn.Dropout(p > 0) in training mode (dnn.py:38) on the fused tile kernel's dropout instance (width 33..64) against
    the generic kernels under the SAME counter-based mask (same seed): loss sums and gradient."""
    r = random.Random(seed)
    res, inn, outn, gc, _, _, _, _ = draw(seed)
    L, W, N = r.choice([1, 2, 4, 8, 11]), r.choice([33, 40, 48, 57, 64]), r.choice([15, 243, 700, 4097, 20000, 70001])
    p = r.choice([0.05, 0.1, 0.3, 0.5])
    d_in, d_out = len(inn), len(outn)
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    if res == "continuity_only":
        X[:, inn.index("x")] *= 40
    base = NetDesc(d_in, d_out, L, W, gc, dropout_p=p)
    params = init_flat_params(base.layers, "xavier", g).cuda()
    nP = base.n_params
    params[nP - d_out:] = torch.rand(d_out, generator=g).cuda() * 0.2
    if res == "physics_equation":
        params[nP - d_out + outn.index("h")] = 0.75
        params[nP - d_out + outn.index("k")] = 0.5
        params[nP - d_out - W * d_out:nP - d_out] *= 0.25     # keep eta_mean + h away from 0: 1 / (rho (eta_mean + h)) is singular there
    spec = ResidualSpec.from_names(res, inn, gc, outn)
    scale = torch.full((spec.n_terms,), 1.0 / N, device="cuda")
    out = {}
    for tag, e in (("generic", ENGINE_GENERIC), ("fused", ENGINE_FUSED_TILE), ("auto", ENGINE_AUTO)):
        eng, grad = Engine(base.with_(engine=e)), torch.zeros(nP, device="cuda")
        eng.dropout_seed = 1000 + seed
        try:
            s = eng.residual_loss_grad(spec, scale, params, X, grad)
        except PinnError as err:      # the dropout instance keeps the gradient copy in LDS: deeper than ~10 x 64 is refused
            assert tag == "fused" and "generic engine" in str(err), err
            continue
        torch.cuda.synchronize()
        out[tag] = (s.double().cpu(), grad.double().cpu())
    s0, g0 = out.pop("generic")
    for tag, (s1, g1) in out.items():
        rel = float((g1 - g0).norm() / g0.norm())
        assert torch.allclose(s1, s0, rtol=1e-4, atol=1e-9), (tag, s1, s0)
        assert rel < 1e-4, (tag, rel)
    print(f"seed {seed}: {res} {d_in}->{L}x{W}->{d_out} k={len(gc)} N={N} p={p}: {'/'.join(out)}")
    assert "auto" in out


@pytest.mark.parametrize("seed", range(9000, 9020))
def test_random_jet_backward_against_the_oracle(seed):
    """pinn_jet_backward (the VJP behind compute_gradient's generic consumers: grad += d/dtheta [sum(gY*Y) + sum(gdY*dY)])
    on random networks, tanh and LeakyReLU, any width up to 300, against torch autograd over the oracle's jet (float64)."""
    from oracle import pinn_oracle as O        # checker only
    d_in, d_out, gc, L, W, N = draw_net(seed)
    r = random.Random(seed)
    if not gc:
        gc = (r.randrange(d_in),)
    W = r.choice([W, 300]) if seed % 5 == 0 else W
    N = min(N, 700)
    init = r.choice(["xavier", "xavier", "kaiming"])
    g = torch.Generator().manual_seed(seed)
    params = O.init_params(O.layer_sizes(d_in, L, W, d_out), init, g)
    X = torch.rand(N, d_in, generator=g) * 2 - 1
    gY, gdY = torch.randn(N, d_out, generator=g), torch.randn(len(gc), N, d_out, generator=g)
    desc = NetDesc(d_in, d_out, L, W, gc, activation=1 if init == "kaiming" else 0)
    eng = Engine(desc)
    for with_gdY in (True, False):
        p64 = [p.double().clone().requires_grad_(True) for p in params]
        cols = O.split_columns(X.double(), gc)
        Yo = O.mlp_forward(p64, torch.cat(cols, -1), init)
        dYo = torch.stack([torch.cat([O.compute_gradient(Yo[:, c:c + 1], cols[j]) for c in range(d_out)], 1) for j in gc])
        obj = (gY.double() * Yo).sum() + ((gdY.double() * dYo).sum() if with_gdY else 0.0)
        ref = O.flat_grad(obj, p64)
        grad = torch.zeros(desc.n_params, device="cuda")
        eng.jet_backward(O.flatten(params).cuda(), X.cuda(), gY.cuda(), gdY.cuda() if with_gdY else None, grad)
        rel = float((grad.double().cpu() - ref).norm() / ref.norm())
        assert rel < 2e-5, (with_gdY, rel)
    print(f"seed {seed}: {d_in}->{L}x{W}->{d_out} {init} k={len(gc)} N={N}: {rel:.1e}")
