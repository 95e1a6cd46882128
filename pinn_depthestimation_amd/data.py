"""data — host-side ingest of the reference's inputs: what train.py:209-277,
train_newmethod.py:216-255 and test.py:150-190 do before the hot path starts.  NumPy / pandas /
SciPy on the host, once per run; nothing here touches the GPU.

Kept from the reference: CSV values rounded to 3 decimals (train.py:218); x / y normalised with
the config's data_test bounds and every other variable with its own nan-min / nan-max
(operations.py:16-30) onto [-1, 1]; a seeded random subset of `training_points` fidelity rows
(np.random.seed(1234) at train.py:22, np.random.choice(..., replace=False) at :238); residual
grids subsampled with [::interval_x, ::interval_y] (:260), flattened COLUMN-major
(reshape -> transpose -> reshape(-1,1), :265-267), stacked as columns, rows containing a NaN
dropped (:276-277).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import operations as op


def column_major_flat(grid: np.ndarray) -> np.ndarray:
    """train.py:265-267"""
    g = np.asarray(grid)
    g = g.reshape(-1, g.shape[1])
    return np.transpose(g).reshape(-1, 1)


def fidelity_from_table(columns: Dict[str, np.ndarray], inputs: Sequence[str], outputs: Sequence[str], config: dict,
                        training_points: Optional[int] = None, rng: Optional[np.random.RandomState] = None):
    """Dict of equally long 1-D columns -> (fidelity_input_train, fidelity_true_train, input_min_max).
    train.py:213-240."""
    fin = {k: np.asarray(columns[k], dtype=np.float64) for k in inputs}
    ftrue = {k: np.asarray(columns[k], dtype=np.float64) for k in outputs}
    mm = op.get_min_max(fin, config)
    for k in fin:
        fin[k] = op.normalize(fin[k], mm[k][0], mm[k][1])
    X = np.column_stack([fin[k] for k in inputs])
    T = np.column_stack([ftrue[k] for k in outputs])
    if training_points is not None:
        chooser = rng if rng is not None else np.random
        idx = chooser.choice(X.shape[0], training_points, replace=False)
        X, T = X[idx, :], T[idx, :]
    return X, T, mm


def load_fidelity_csv(path: str, config: dict, rng: Optional[np.random.RandomState] = None):
    """train.py:209-240 for config['data_fidelity'] (file, inputs, outputs, training_points)."""
    import pandas as pd
    df = config["data_fidelity"]
    table = pd.read_csv(path).round(3)
    cols = {k: table[k].to_numpy() for k in table.columns}
    return fidelity_from_table(cols, df["inputs"], df["outputs"], config, df.get("training_points"), rng)


def residual_from_grids(grids: Dict[str, np.ndarray], inputs: Sequence[str], input_min_max: Dict[str, Tuple[float, float]],
                        interval_x: int = 1, interval_y: int = 1) -> np.ndarray:
    """Dict of 2-D grids -> (N, len(inputs)) collocation matrix.  train.py:257-277."""
    out = None
    for k in inputs:
        g = np.asarray(grids[k])[::interval_x, ::interval_y]
        g = op.normalize(g, input_min_max[k][0], input_min_max[k][1])
        col = column_major_flat(g)
        out = col if out is None else np.hstack((out, col))
    return out[~np.isnan(out).any(axis=1)]


def load_residual_mat(path: str, config: dict, input_min_max: Optional[dict] = None) -> np.ndarray:
    """train.py:246-277 for config['data_residual'] (file, inputs, interval_x, interval_y)."""
    from scipy.io import loadmat
    dr = config["data_residual"]
    names = list(dr["inputs"].keys()) if isinstance(dr["inputs"], dict) else list(dr["inputs"])
    grids = {k: loadmat(path, variable_names=k)[k] for k in names}
    if input_min_max is None:
        input_min_max = op.get_min_max(grids, config)
    return residual_from_grids(grids, names, input_min_max, dr.get("interval_x", 1), dr.get("interval_y", 1))


def grid_inputs(grids: Dict[str, np.ndarray], inputs: Sequence[str], config: dict):
    """test.py:156-181: normalise whole grids and flatten ROW-major into an (ny*nx, d_in) matrix."""
    mm = op.get_min_max(grids, config)
    cols = [op.normalize(np.asarray(grids[k]), mm[k][0], mm[k][1]).reshape(-1, 1) for k in inputs]
    return np.hstack(cols), mm


# ---- the same staging on the device ("data ingest + normalisation on device") -------------------
def _lib_and_stream(device):
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return lib, C, torch, torch.device("cuda", idx), C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


def device_nan_min_max(t):
    """np.nanmin / np.nanmax of a float64 device tensor (operations.py:26-27) -> (2,) float64 device tensor,
    no host round trip (pinn_nanminmax_f64)."""
    lib, C, torch, dev, stream = _lib_and_stream(t.device)
    from ._lib import check
    t = t.contiguous()
    if t.dtype != torch.float64:
        raise ValueError("device_nan_min_max takes the float64 array loadmat produced")
    out = torch.empty(2, dtype=torch.float64, device=dev)
    ws = torch.empty(1024 * 16, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.pinn_nanminmax_f64(C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(ws.data_ptr()), ws.numel(), stream), "pinn_nanminmax_f64")
    return out


def stage_residual_on_device(grids: Dict[str, np.ndarray], inputs: Sequence[str], config: Optional[dict] = None,
                             input_min_max: Optional[Dict[str, Tuple[float, float]]] = None, interval_x: int = 1,
                             interval_y: int = 1, device="cuda"):
    """train.py:246-277 with the grids on the GPU: dict of (ny, nx) float64 grids (host arrays are uploaded as they
    are, device tensors are used in place) -> (N, len(inputs)) float32 collocation matrix ON THE DEVICE, equal bit
    for bit to torch.tensor(residual_from_grids(...)).float() (pinn_stage_grid_columns: subsample, min-max normalise,
    column-major flatten, NaN-row compaction, fp32 cast).  Bounds: `input_min_max[k]` where given (train.py reuses
    the fidelity table's), else x / y from config['data_test'] and every other variable's own nan-min / nan-max,
    reduced on the device (operations.py:16-30).  One synchronisation (the number of surviving rows)."""
    lib, C, torch, dev, stream = _lib_and_stream(device)
    from ._lib import check
    g = []
    for k in inputs:
        t = grids[k] if torch.is_tensor(grids[k]) else torch.from_numpy(np.ascontiguousarray(np.asarray(grids[k], dtype=np.float64)))
        g.append(t.to(device=dev, dtype=torch.float64).contiguous())
    ny, nx = g[0].shape
    if any(tuple(t.shape) != (ny, nx) for t in g):
        raise ValueError("all input grids must have the same (ny, nx) shape")
    mm = torch.empty(len(inputs), 2, dtype=torch.float64, device=dev)
    for c, k in enumerate(inputs):
        if input_min_max is not None and k in input_min_max:
            mm[c] = torch.tensor([float(input_min_max[k][0]), float(input_min_max[k][1])], dtype=torch.float64)
        elif k in ("x", "y"):
            mm[c] = torch.tensor([float(config["data_test"][f"{k}_min"]), float(config["data_test"][f"{k}_max"])], dtype=torch.float64)
        else:
            mm[c] = device_nan_min_max(g[c])
    rows = -(-ny // interval_x) * -(-nx // interval_y)
    X = torch.empty(rows, len(inputs), dtype=torch.float32, device=dev)
    n_rows = torch.zeros(1, dtype=torch.int64, device=dev)
    need = lib.pinn_stage_workspace_bytes(ny, nx, interval_x, interval_y)
    ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
    ptrs = (C.c_void_p * len(g))(*[t.data_ptr() for t in g])
    with torch.cuda.device(dev):
        check(lib.pinn_stage_grid_columns(ptrs, len(g), ny, nx, interval_x, interval_y, C.c_void_p(mm.data_ptr()),
                                          C.c_void_p(X.data_ptr()), C.c_void_p(n_rows.data_ptr()),
                                          C.c_void_p(ws.data_ptr()), ws.numel(), stream), "pinn_stage_grid_columns")
    return X[:int(n_rows.item())], mm
