"""run — the reference's script entry points as functions: train.py:203-288 (`python train.py`)
and test.py:136-192 (`python test.py`), on the HIP engine.

    python -m pinn_depthestimation_amd.run train config_CMB.json [log_dir]
    python -m pinn_depthestimation_amd.run test  config_CMB.json model.pth

File formats, normalisation, subsampling, seeding, artefacts are the reference's (see data.py,
trainer.py, inference.py); the config file is read verbatim.  Paths inside the config are taken
relative to the current directory, as the reference does.
"""
from __future__ import annotations

import datetime
import json
import os
import sys
import time
from typing import Optional

import numpy as np
import torch

from . import data as D
from .inference import Tester
from .trainer import PINN


def _raw(config):
    return config if isinstance(config, dict) else json.load(open(config, "r"))


def train_from_config(config, log_dir: Optional[str] = None, device="cuda", **pinn_kwargs) -> PINN:
    cfg = _raw(config)
    np.random.seed(1234)                                              # train.py:22
    if log_dir is None:
        log_dir = f"../log/{datetime.datetime.now().strftime('%Y%m%d_%H%M')}"   # train.py:39-43
    os.makedirs(log_dir, exist_ok=True)
    Xf, Tf, input_min_max = D.load_fidelity_csv(cfg["data_fidelity"]["file"], cfg)       # train.py:209-240
    Xr = D.load_residual_mat(cfg["data_residual"]["file"], cfg, input_min_max)           # train.py:246-277
    model = PINN(Xf, Tf, Xr, cfg, device=device, log_dir=log_dir, **pinn_kwargs)         # train.py:280
    t0 = time.time()
    model.train()                                                                       # train.py:284
    if model.reducer.rank == 0:
        print("Training time: %.4f" % (time.time() - t0))                               # train.py:286
    model.save_checkpoint("model.pth")                                                   # train.py:288
    return model


def test_from_config(config, model_path: Optional[str] = None, device="cuda", timesteps=(0,)):
    """test.py:136-192: load the grids named in config['data_test'], normalise, predict on the full grid."""
    from scipy.io import loadmat
    cfg = _raw(config)
    dt = cfg["data_test"]
    tester = Tester(model_path or dt["model"], cfg, device=device)
    names = list(dt["inputs"].keys()) if isinstance(dt["inputs"], dict) else list(dt["inputs"])
    grids = {k: loadmat(dt["file"], variable_names=k)[k] for k in names}
    X, input_min_max = D.grid_inputs(grids, names, cfg)
    preds = []
    for i in timesteps:
        preds.append(tester.test(X, input_min_max=input_min_max))
        print(f"Done: Prediction for timestep: {i}")                                    # test.py:192
    return tester, preds


if __name__ == "__main__":
    if len(sys.argv) < 3 or sys.argv[1] not in ("train", "test"):
        print(__doc__)
        sys.exit(2)
    if sys.argv[1] == "train":
        train_from_config(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        test_from_config(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
