"""Helpers to read tests/golden/*.npz (written by tests/golden/make_goldens.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def state_dict(z, prefix="sd/"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def layers_of(sd):
    n = len([k for k in sd if k.endswith(".weight")])
    return [sd["layers.layer_0.weight"].shape[1]] + [sd[f"layers.layer_{i}.weight"].shape[0] for i in range(n)]


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().reshape(-1), torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-300))
