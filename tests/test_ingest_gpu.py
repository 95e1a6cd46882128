"""Device-side staging of the collocation points (pinn_stage_grid_columns / pinn_nanminmax_f64)
against the host NumPy path (pinn_depthestimation_amd/data.py, itself checked against literal restatements of
train.py:246-277 in tests/test_data_cpu.py): BIT-identical fp32 matrices — same float64 arithmetic, same order,
fp32 cast last."""
import numpy as np
import pytest
import torch

from pinn_depthestimation_amd import data as D

pytestmark = pytest.mark.gpu

CFG = {"data_test": {"x_min": 25.0, "x_max": 33.0, "y_min": -13.0, "y_max": 13.0}}


def grids(ny, nx, seed, nan_frac):
    rng = np.random.RandomState(seed)
    xg, yg = np.meshgrid(np.linspace(25, 33, nx), np.linspace(-13, 13, ny))
    t = rng.uniform(0, 90, (ny, nx))
    h = rng.uniform(0.7, 0.8, (ny, nx)) + 1e-3 * rng.randn(ny, nx)
    h[rng.rand(ny, nx) < nan_frac] = np.nan                       # dry cells / masked pixels
    xg = xg.copy(); xg[rng.rand(ny, nx) < nan_frac / 2] = np.nan
    return {"x": xg, "y": yg, "t": t, "h": h}


@pytest.mark.parametrize("ny,nx,ix,iy,nan_frac", [(9, 7, 2, 3, 0.1), (81, 261, 1, 1, 0.02), (300, 517, 3, 2, 0.3),
                                                  (1, 1, 1, 1, 0.0), (64, 64, 5, 7, 1.0), (257, 1000, 1, 1, 0.0)])
def test_device_staging_is_bit_identical_to_the_host_path(ny, nx, ix, iy, nan_frac):
    g = grids(ny, nx, ny * 1000 + nx, nan_frac)
    names = ["t", "x", "y", "h"]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                           # np.nanmin of an all-NaN slice warns, as in the reference
        mm = D.op.get_min_max({k: g[k] for k in names}, CFG)      # train.py:228 call form
        host = D.residual_from_grids(g, names, mm, ix, iy)
    want = torch.tensor(host).float()                             # train.py:88
    X, mm_dev = D.stage_residual_on_device(g, names, CFG, None, ix, iy)
    assert X.dtype == torch.float32 and X.is_cuda and tuple(X.shape) == tuple(want.shape)
    assert torch.equal(X.cpu(), want)
    for c, k in enumerate(names):                                 # the device's own nan-min / nan-max reductions
        lo, hi = mm_dev[c].tolist()
        assert (np.isnan(lo) and np.isnan(mm[k][0])) or lo == float(mm[k][0])
        assert (np.isnan(hi) and np.isnan(mm[k][1])) or hi == float(mm[k][1])
    # bounds handed in (train.py reuses the fidelity table's input_min_max) and a degenerate range -> zeros
    mm2 = {"t": (10.0, 80.0), "x": (25.0, 33.0), "y": (-13.0, 13.0), "h": (0.75, 0.75)}
    host2 = torch.tensor(D.residual_from_grids(g, names, mm2, ix, iy)).float()
    X2, _ = D.stage_residual_on_device(g, names, CFG, mm2, ix, iy)
    assert torch.equal(X2.cpu(), host2)


def test_staged_points_feed_the_engine():
    """The staged matrix goes straight into the hot path: forward on it equals forward on the host-staged one."""
    from pinn_depthestimation_amd import Engine, NetDesc
    from pinn_depthestimation_amd.dnn import init_flat_params
    g = grids(81, 261, 5, 0.05)
    X, _ = D.stage_residual_on_device(g, ["t", "x", "y"], CFG)
    host = torch.tensor(D.residual_from_grids(g, ["t", "x", "y"], D.op.get_min_max({k: g[k] for k in "txy"}, CFG))).float().cuda()
    desc = NetDesc(3, 4, 8, 64, (0, 1, 2))
    flat = init_flat_params(desc.layers, "xavier", torch.Generator().manual_seed(0)).cuda()
    eng = Engine(desc)
    assert torch.equal(eng.forward(flat, X.contiguous()), eng.forward(flat, host))


@pytest.mark.parametrize("seed", range(40))
def test_random_grids_stage_bit_identically(seed):
    """Seeded sweep over grid shape, sub-sampling strides, NaN density (none .. everything), variable count and order:
    the device path equals the host NumPy path bit for bit."""
    import warnings
    rng = np.random.RandomState(seed)
    ny, nx = int(rng.choice([1, 2, 3, 17, 64, 129, 300])), int(rng.choice([1, 2, 5, 33, 256, 517, 1031]))
    ix, iy = int(rng.randint(1, 6)), int(rng.randint(1, 6))
    nan_frac = float(rng.choice([0.0, 0.0, 0.01, 0.2, 0.9, 1.0]))
    g = grids(ny, nx, 77 + seed, nan_frac)
    names = list(rng.permutation(["t", "x", "y", "h"])[:rng.randint(1, 5)])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mm = D.op.get_min_max({k: g[k] for k in names}, CFG)
        host = D.residual_from_grids(g, names, mm, ix, iy)
    want = torch.tensor(host).float()
    X, mm_dev = D.stage_residual_on_device(g, names, CFG, None, ix, iy)
    assert tuple(X.shape) == tuple(want.shape), (X.shape, want.shape)
    assert torch.equal(X.cpu(), want)
    for c, k in enumerate(names):
        lo, hi = mm_dev[c].tolist()
        assert (np.isnan(lo) and np.isnan(mm[k][0])) or lo == float(mm[k][0])
        assert (np.isnan(hi) and np.isnan(mm[k][1])) or hi == float(mm[k][1])
