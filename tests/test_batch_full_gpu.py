"""The batch kernel's FULL-batch instances (fused_batch_kernel.h: T = 4 / 2 tiles per wave and batch at width <= 32, 4 / 2
with two waves per SIMD at width <= 16) only run once every wave of the chip has a batch — above 65 536 / 131 072 points;
the oracle-sized cases of test_engine_gpu.py reach the one-tile instances only.  Here each full-batch instance family is
compared with the tile kernel (k_fused, itself pinned by the oracle) on the same 2^18 points: same loss sums, same
gradient, for both gradient sinks (per-wave LDS copies / global atomics)."""
import pytest
import torch

from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd._lib import ENGINE_FUSED_BATCH, ENGINE_FUSED_TILE
from pinn_depthestimation_amd.dnn import init_flat_params

pytestmark = pytest.mark.gpu

SHAPES = {
    # name: d_in, d_out, L, W, grad_cols, residual, inputs, outputs
    "pe10x10 (width 16, k=2, LDS sink)": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    "ns6x12 (width 16, k=3, LDS sink)": (3, 4, 6, 12, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "co60x20 (width 32, k=2, atomic sink)": (2, 3, 60, 20, (0, 1), "continuity_only", ("x", "y"), ("U", "V", "h")),
    "ns20x20 (width 32, k=3, atomic sink)": (4, 4, 20, 20, (0, 1, 2), "Navier_Stokes", ("t", "x", "y", "z0"), ("h", "z", "u", "v")),
    "cf3x28 (width 32, eight k-steps)": (2, 3, 3, 28, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h")),
}


@pytest.mark.parametrize("name", list(SHAPES))
def test_full_batch_instances_match_the_tile_kernel(name):
    d_in, d_out, L, W, gc, res, inn, outn = SHAPES[name]
    N = 1 << 18
    X = (torch.rand(N, d_in, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
    if res == "continuity_only":
        X[:, 0] *= 40            # x < 25.5 selects a real subset
    out = {}
    for tag, e in (("tile", ENGINE_FUSED_TILE), ("batch", ENGINE_FUSED_BATCH)):
        desc = NetDesc(d_in, d_out, L, W, gc, engine=e)
        spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
        params = init_flat_params(desc.layers, "xavier", torch.Generator().manual_seed(3)).cuda()
        if res == "physics_equation":
            params[desc.n_params - d_out + 0] = 0.75
            params[desc.n_params - d_out + 3] = 0.0
        grad = torch.zeros(desc.n_params, device="cuda")
        scale = torch.full((spec.n_terms,), 1.0 / N, device="cuda")
        sums = Engine(desc).residual_loss_grad(spec, scale, params, X, grad)
        torch.cuda.synchronize()
        out[tag] = (sums.clone(), grad.clone())
    (s0, g0), (s1, g1) = out["tile"], out["batch"]
    assert torch.allclose(s0, s1, rtol=2e-5), (s0, s1)
    rel = float((g0 - g1).norm() / g0.norm())
    print(f"{name}: gradient rel-L2 difference tile vs batch {rel:.2e}")
    assert rel < 5e-6


@pytest.mark.parametrize("shape", [(3, 4, 8, 64, 0), (2, 6, 3, 48, 0), (5, 9, 1, 33, 0), (3, 4, 5, 64, 1),
                                   (2, 6, 10, 10, 0), (2, 3, 100, 20, 0), (4, 4, 20, 20, 0), (3, 5, 2, 7, 1), (1, 1, 1, 32, 0)])
def test_plain_forward_four_tiles_per_wave_is_bitwise_the_tile_kernel(shape):
    """pinn_forward above 131 072 points at padded width 64 (262 144 at 16 / 32) runs k_fused_plain4
    (pinn_fused_plain.hip: four tiles per wave share each weight fetch).  Same fmaf chain per point as the one-tile kernel: Y must be bit-identical
    (ragged point count: the last pass has one live tile and a half-empty one), and it must agree with the generic kernels."""
    from pinn_depthestimation_amd._lib import ENGINE_AUTO, ENGINE_GENERIC
    d_in, d_out, L, W, act = shape
    N = 150001 if W > 32 else 300001
    g = torch.Generator().manual_seed(21)
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    desc = NetDesc(d_in, d_out, L, W, (), activation=act)
    params = init_flat_params(desc.layers, "kaiming" if act else "xavier", g).cuda()
    Y = {tag: Engine(desc.with_(engine=e)).forward(params, X) for tag, e in
         (("auto", ENGINE_AUTO), ("tile", ENGINE_FUSED_TILE), ("generic", ENGINE_GENERIC))}
    torch.cuda.synchronize()
    assert torch.equal(Y["auto"], Y["tile"])
    assert float((Y["auto"] - Y["generic"]).abs().max()) < 5e-6 * max(1.0, float(Y["generic"].abs().max()))
    # a differentiated column in the descriptor does not change what pinn_forward runs (it drops the tangents)
    jet = NetDesc(d_in, d_out, L, W, (0,), activation=act)
    assert torch.equal(Engine(jet).forward(params, X), Y["auto"])
