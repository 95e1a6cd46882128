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
