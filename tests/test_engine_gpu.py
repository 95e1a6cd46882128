"""Parity of the C-ABI engine against the CPU oracle on seeded inputs (GPU only).

Tolerances (fp32): forward / jet 2e-6 abs; loss sums 2e-6 rel against an fp64 run of the
oracle formulation; flat gradient 2e-5 relative L2.  The reference's own fp32-vs-fp64
noise floor is 8.6e-8 (loss) / 1.4e-7 (grad) for Navier_Stokes.
"""
import pytest
import torch

from oracle import pinn_oracle as O
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd._lib import (ACT_LEAKY_RELU, ACT_TANH, ENGINE_FUSED, ENGINE_FUSED_BATCH, ENGINE_FUSED_COOP,
                                           ENGINE_FUSED_TILE, ENGINE_GENERIC)

from tests.golden_util import oracle_loss_and_grad, rel_l2

pytestmark = pytest.mark.gpu

CASES = {
    # name: (d_in, d_out, hidden, width, grad_cols, residual, in names, out names)
    "ns_8x64": (3, 4, 8, 64, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "ns_5in": (5, 4, 3, 20, (0, 1, 2), "Navier_Stokes", ("t", "x", "y", "u0", "v0"), ("h", "z", "u", "v")),
    "pe_10x10": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    "pe_8x64": (2, 6, 8, 64, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    "cf_4x20": (2, 3, 4, 20, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h")),
    "co_4x20": (2, 3, 4, 20, (0, 1), "continuity_only", ("x", "y"), ("U", "V", "h")),
    # padded width 64 (where the tile / cooperative / paired kernels differ): shallow nets, every residual family
    "ns_1x64": (3, 4, 1, 64, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "ns_2x48": (3, 4, 2, 48, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "co_3x64": (2, 3, 3, 64, (0, 1), "continuity_only", ("x", "y"), ("U", "V", "h")),
    # deep and narrow (three of the reference's four configs are 20-100 layers of width 20): the padded
    # gradient no longer fits LDS and is accumulated in per-workgroup global copies
    "cf_40x20": (2, 3, 40, 20, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h")),
    # corners of the batch kernel's instance table (fused_batch_kernel.h): k = 3 at width <= 16 (two tiles per batch),
    # four k-steps at width 16, eight at width 32, a single hidden layer (no hidden-to-hidden matrix at all)
    "ns_3x12": (3, 4, 3, 12, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "pe_2x16": (2, 6, 2, 16, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    "cf_3x32": (2, 3, 3, 32, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h")),
    "pe_1x10": (2, 6, 1, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    # output column 0 is NOT one of the residual's roles, roles spill into the second group of four output columns
    # (round-3 regression: unused spec entries claimed column 0; found by test_sweep_gpu.py)
    "ns_out_first_3x12": (3, 7, 3, 12, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("aux0", "aux1", "h", "z", "u", "v", "aux2")),
    "cf_out_first_2x64": (2, 5, 2, 64, (0, 1), "continuity_ftemp", ("x", "y"), ("aux0", "aux1", "h", "U", "V")),
}


def make_case(name, N, seed=1234, dtype=torch.float32):
    d_in, d_out, L, W, gc, res, inn, outn = CASES[name]
    g = torch.Generator().manual_seed(seed)
    layers = O.layer_sizes(d_in, L, W, d_out)
    params = O.init_params(layers, "xavier", g)
    if res == "physics_equation":
        # keep eta_mean + h away from 0 (1/(rho*(eta+h)) is singular there)
        params[-1][outn.index("h")] = 0.75
        params[-1][outn.index("eta_mean")] = 0.0
    X = (torch.rand(N, d_in, generator=g) * 2 - 1)
    return layers, params, X, NetDesc(d_in, d_out, L, W, gc), res, inn, outn


ENGINES = [ENGINE_GENERIC, ENGINE_FUSED]


@pytest.fixture(autouse=True, params=["tile", "coop", "batch"])
def fused_kernel_choice(request, monkeypatch):
    """The fused engine has two kernels at padded width 64: one wave per 16-point tile (k_fused) and four
    waves per tile (k_fused_coop, picked automatically for small N); networks of width <= 32 have the tile kernel
    and the batch kernel (k_fused_batch: layer-major batches of 8 tiles per wave, picked automatically for large
    N; gradient passes only — its forward-only requests run on the tile kernel).  Every test in this module runs
    with each forced in turn through the descriptor (pinn_desc.engine = PINN_ENGINE_FUSED_TILE / _COOP / _BATCH —
    the library reads no environment variables)."""
    orig = NetDesc.c_struct

    def c_struct(self):
        d = orig(self)
        if d.engine == ENGINE_FUSED:
            coop = request.param == "coop" and 32 < self.width <= 64
            batch = request.param == "batch" and self.width <= 32
            d.engine = ENGINE_FUSED_COOP if coop else (ENGINE_FUSED_BATCH if batch else ENGINE_FUSED_TILE)
        return d
    monkeypatch.setattr(NetDesc, "c_struct", c_struct)
    return request.param


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name", ["ns_8x64", "ns_5in", "pe_10x10", "cf_4x20"])
def test_forward_and_jet(name, engine):
    layers, params, X, desc, res, inn, outn = make_case(name, 300)
    eng = Engine(desc.with_(engine=engine))
    flat = O.flatten(params).cuda()
    Xd = X.cuda().contiguous()
    Y = eng.forward(flat, Xd)
    Yj, dY = eng.forward_jet(flat, Xd)
    Yo, dYo = O.jet([p.double() for p in params], X.double(), desc.grad_cols)
    assert (Y.cpu().double() - Yo).abs().max() < 2e-6
    assert (Yj.cpu().double() - Yo).abs().max() < 2e-6
    assert (dY.cpu().double() - dYo).abs().max() < 2e-6 * max(1.0, float(dYo.abs().max()))


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name", list(CASES))
def test_residual_loss_grad(name, engine):
    N = 777  # ragged: not a multiple of any tile size
    layers, params, X, desc, res, inn, outn = make_case(name, N)
    if res == "continuity_only":
        X[:, 0] = X[:, 0] * 40  # make x < 25.5 a real subset
    eng = Engine(desc.with_(engine=engine))
    spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
    flat = O.flatten(params).cuda()
    Xd = X.cuda().contiguous()
    l64, g64 = oracle_loss_and_grad(params, X, res, inn, outn, desc.grad_cols, torch.float64)
    l32, g32 = oracle_loss_and_grad(params, X, res, inn, outn, desc.grad_cols, torch.float32)
    if res == "continuity_only":
        cnt = float((X[:, 0] < 25.5).sum())
        scale = torch.tensor([1.0 / N, 1.0 / cnt, 0.0])
    else:
        scale = torch.full((spec.n_terms,), 1.0 / N)
    grad = torch.zeros(desc.n_params, device="cuda")
    sums = eng.residual_loss_grad(spec, scale.cuda(), flat, Xd, grad)
    sums_only = eng.residual_loss(spec, flat, Xd)
    loss = float((sums.cpu().double() * scale.double()).sum())
    assert torch.allclose(sums, sums_only, rtol=1e-6, atol=0)
    if res == "continuity_only":
        assert float(sums[2]) == cnt
    ref_noise = abs(float(l32) - float(l64)) / abs(float(l64))
    assert abs(loss - float(l64)) / abs(float(l64)) < max(2e-6, 4 * ref_noise)
    gnoise = rel_l2(g32, g64)
    assert rel_l2(grad.cpu(), g64) < max(2e-5, 4 * gnoise)


@pytest.mark.parametrize("case", ["pe_10x10", "pe_8x64"])
@pytest.mark.parametrize("engine", ENGINES)
def test_mse_loss_grad(engine, case):
    layers, params, X, desc, *_ = make_case(case, 12)
    g = torch.Generator().manual_seed(7)
    T = torch.rand(12, 6, generator=g)
    w = [1.0, 2.0, 0.5, 1.0, 3.0, 1.0]
    p = [q.double().requires_grad_(True) for q in params]
    lo = O.fidelity_loss(p, X.double(), T.double(), list(range(6)), w)
    go = O.flat_grad(lo, p)
    eng = Engine(desc.with_(engine=engine))
    grad = torch.zeros(desc.n_params, device="cuda")
    scale = torch.tensor(w) / 12
    sums = eng.mse_loss_grad(O.flatten(params).cuda(), X.cuda(), T.cuda(), list(range(6)), scale.cuda(), grad)
    loss = float((sums.cpu().double() * scale.double()).sum())
    assert abs(loss - float(lo)) / float(lo) < 2e-6
    assert rel_l2(grad.cpu(), go) < 2e-5


@pytest.mark.parametrize("engine", ENGINES)
def test_jet_backward_matches_autograd(engine):
    layers, params, X, desc, *_ = make_case("ns_5in", 200)
    g = torch.Generator().manual_seed(3)
    gY = torch.randn(200, 4, generator=g)
    gdY = torch.randn(3, 200, 4, generator=g)
    p = [q.double().requires_grad_(True) for q in params]
    cols = O.split_columns(X.double(), desc.grad_cols)
    Y = O.mlp_forward(p, torch.cat(cols, -1))
    dY = torch.stack([torch.cat([O.compute_gradient(Y[:, c:c + 1], cols[j]) for c in range(4)], 1) for j in desc.grad_cols])
    obj = (Y * gY.double()).sum() + (dY * gdY.double()).sum()
    go = O.flat_grad(obj, p)
    eng = Engine(desc.with_(engine=engine))
    grad = torch.zeros(desc.n_params, device="cuda")
    eng.jet_backward(O.flatten(params).cuda(), X.cuda(), gY.cuda(), gdY.cuda(), grad)
    assert rel_l2(grad.cpu(), go) < 2e-5


@pytest.mark.parametrize("engine", ENGINES)
def test_leaky_relu_kaiming_network(engine):
    """init_type 'kaiming' -> LeakyReLU(0.01) (dnn.py:20-21): jet, loss and gradient on both engines."""
    d = NetDesc(2, 3, 3, 16, (0, 1), ACT_LEAKY_RELU, engine)
    g = torch.Generator().manual_seed(5)
    params = O.init_params(d.layers, "kaiming", g)
    X = torch.rand(333, 2, generator=g) * 2 - 1
    eng = Engine(d)
    flat = O.flatten(params).cuda()
    Y, dY = eng.forward_jet(flat, X.cuda())
    Yo, dYo = O.jet([p.double() for p in params], X.double(), (0, 1), "kaiming")
    assert (Y.cpu().double() - Yo).abs().max() < 2e-6
    assert (dY.cpu().double() - dYo).abs().max() < 5e-6
    p64 = [p.double().requires_grad_(True) for p in params]
    lo = O.residual_loss(p64, X.double(), "continuity_ftemp", [0, 1], [2, 0, 1], (0, 1), "kaiming")
    go = O.flat_grad(lo, p64)
    spec = ResidualSpec.from_names("continuity_ftemp", ("x", "y"), (0, 1), ("U", "V", "h"))
    grad = torch.zeros(d.n_params, device="cuda")
    scale = torch.full((1,), 1.0 / 333, device="cuda")
    sums = eng.residual_loss_grad(spec, scale, flat, X.cuda(), grad)
    assert abs(float(sums[0]) / 333 - float(lo)) / float(lo) < 5e-6
    assert rel_l2(grad.cpu(), go) < 2e-5


def test_adam_step_matches_torch():
    g = torch.Generator().manual_seed(11)
    eng = Engine(NetDesc(2, 3, 2, 4, (0,)))
    P = eng.n_params
    p0 = torch.randn(P, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-4)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.8)
    pd, m, v = p0.cuda(), torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    for step in range(1, 9):
        gr = torch.randn(P, generator=g)
        ref.grad = gr.clone()
        lr = opt.param_groups[0]["lr"]
        opt.step(); sch.step()
        eng.adam_step(pd, gr.cuda(), m, v, step, lr)
        assert torch.equal(pd.cpu(), ref.detach()), f"step {step}: max diff {(pd.cpu()-ref.detach()).abs().max()}"


@pytest.mark.parametrize("case", ["co_4x20", "co_3x64"])
@pytest.mark.parametrize("engine", ENGINES)
def test_residual_and_mse_in_one_pass(engine, case):
    """pinn_residual_mse_loss_grad (train_newmethod.py:122-159: one forward feeds both terms) equals the
    two separate calls on the same points."""
    layers, params, X, desc, res, inn, outn = make_case(case, 555)
    X[:, 0] = X[:, 0] * 40
    g = torch.Generator().manual_seed(8)
    T = torch.rand(555, 2, generator=g).cuda()
    eng = Engine(desc.with_(engine=engine))
    spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
    flat, Xd = O.flatten(params).cuda(), X.cuda().contiguous()
    cnt = float((X[:, 0] < 25.5).sum())
    ts = torch.tensor([1.0 / 555, 1.0 / cnt, 0.0]).cuda()
    cs = torch.tensor([0.7 / 555, 1.3 / 555]).cuda()
    g1 = torch.zeros(desc.n_params, device="cuda")
    s_res = eng.residual_loss_grad(spec, ts, flat, Xd, g1)
    s_mse = eng.mse_loss_grad(flat, Xd, T, [0, 1], cs, g1)
    g2 = torch.zeros(desc.n_params, device="cuda")
    t_sums, c_sums = eng.residual_mse_loss_grad(spec, ts, T, [0, 1], cs, flat, Xd, g2)
    assert torch.allclose(t_sums, s_res, rtol=1e-6) and torch.allclose(c_sums, s_mse, rtol=1e-6)
    assert rel_l2(g2.cpu(), g1.cpu()) < 2e-6


@pytest.mark.parametrize("shape", [
    (3, 5, 3, 48, (1,)),          # k = 1 jet at padded width 64
    # (d_in, d_out, hidden layers, width, grad cols): unusual I/O widths, one differentiated input
    # (K1 = 2), a single hidden layer, a non-multiple-of-16 wide net
    (7, 13, 2, 24, (3,)),
    (16, 16, 1, 64, (0, 5, 15)),
    (1, 1, 3, 8, (0,)),
    (4, 9, 2, 100, (2, 3, 0)),
])
def test_forward_and_jet_odd_shapes_all_engines(shape):
    d_in, d_out, L, W, gc = shape
    g = torch.Generator().manual_seed(d_in * 100 + d_out)
    desc = NetDesc(d_in, d_out, L, W, gc)
    params = O.init_params(desc.layers, "xavier", g)
    X = torch.rand(211, d_in, generator=g) * 2 - 1
    Yo, dYo = O.jet([p.double() for p in params], X.double(), gc)
    flat, Xd = O.flatten(params).cuda(), X.cuda()
    for engine in (0, ENGINE_GENERIC):          # AUTO (fused / wide where supported) and the generic engine
        eng = Engine(desc.with_(engine=engine))
        Y, dY = eng.forward_jet(flat, Xd)
        assert (Y.cpu().double() - Yo).abs().max() < 3e-6
        assert (dY.cpu().double() - dYo).abs().max() < 3e-6 * max(1.0, float(dYo.abs().max()))
        assert (eng.forward(flat, Xd).cpu().double() - Yo).abs().max() < 3e-6
        # VJP of the jet (generic consumer path) against autograd
        gY = torch.randn(211, d_out, generator=g)
        gdY = torch.randn(len(gc), 211, d_out, generator=g)
        p64 = [q.double().requires_grad_(True) for q in params]
        cols = O.split_columns(X.double(), gc)
        Yt = O.mlp_forward(p64, torch.cat(cols, -1))
        dYt = torch.stack([torch.cat([O.compute_gradient(Yt[:, c:c + 1], cols[j]) for c in range(d_out)], 1) for j in gc])
        go = O.flat_grad((Yt * gY.double()).sum() + (dYt * gdY.double()).sum(), p64)
        grad = torch.zeros(desc.n_params, device="cuda")
        eng.jet_backward(flat, Xd, gY.cuda(), gdY.cuda(), grad)
        assert rel_l2(grad.cpu(), go) < 3e-5


@pytest.mark.parametrize("case", ["ns_8x64", "pe_10x10", "co_3x64", "cf_4x20"])
@pytest.mark.parametrize("engine", ENGINES)
def test_split_pass_equals_residual_plus_fidelity_calls(engine, case):
    """pinn_residual_mse_split_loss_grad (train.py:131-157 in one launch: collocation points first, fidelity
    points after them) equals pinn_residual_loss_grad on the first set plus pinn_mse_loss_grad on the second."""
    n_res, n_fid = 333, 45          # neither a multiple of a tile; the boundary falls inside a tile
    layers, params, X, desc, res, inn, outn = make_case(case, n_res + n_fid)
    if res == "continuity_only":
        X[:, 0] = X[:, 0] * 40
    nc = min(3, desc.d_out)
    g = torch.Generator().manual_seed(11)
    T = torch.rand(n_fid, nc, generator=g).cuda()
    eng = Engine(desc.with_(engine=engine))
    spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
    flat, Xd = O.flatten(params).cuda(), X.cuda().contiguous()
    ts = torch.full((spec.n_terms,), 1.0 / n_res).cuda()
    if res == "continuity_only":
        ts = torch.tensor([1.0 / n_res, 1.0 / max(float((X[:n_res, 0] < 25.5).sum()), 1.0), 0.0]).cuda()
    cs = (torch.tensor([0.7, 1.3, 2.0][:nc]) / n_fid).cuda()
    cols = list(range(nc))
    g1 = torch.zeros(desc.n_params, device="cuda")
    s_res = eng.residual_loss_grad(spec, ts, flat, Xd[:n_res].contiguous(), g1)
    s_mse = eng.mse_loss_grad(flat, Xd[n_res:].contiguous(), T, cols, cs, g1)
    g2 = torch.zeros(desc.n_params, device="cuda")
    t_sums, c_sums = eng.residual_mse_split_loss_grad(spec, ts, T, cols, cs, flat, Xd, n_res, g2)
    assert torch.allclose(t_sums, s_res, rtol=2e-6), (t_sums, s_res)
    assert torch.allclose(c_sums, s_mse, rtol=2e-6), (c_sums, s_mse)
    assert rel_l2(g2.cpu(), g1.cpu()) < 3e-6
    # degenerate splits: no fidelity points / no collocation points
    g3 = torch.zeros(desc.n_params, device="cuda")
    t3, c3 = eng.residual_mse_split_loss_grad(spec, ts, T[:0], cols, cs, flat, Xd[:n_res].contiguous(), n_res, g3)
    assert torch.allclose(t3, s_res, rtol=2e-6) and float(c3.abs().sum()) == 0.0
    g4 = torch.zeros(desc.n_params, device="cuda")
    t4, c4 = eng.residual_mse_split_loss_grad(spec, ts, T, cols, cs, flat, Xd[n_res:].contiguous(), 0, g4)
    assert torch.allclose(c4, s_mse, rtol=2e-6) and float(t4[:2].abs().sum()) == 0.0
