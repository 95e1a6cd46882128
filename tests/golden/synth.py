"""Deterministic synthetic weights shared by make_goldens.py (which loads them into the REFERENCE's
dnn.DNN) and by the GPU tests (which hand them to the HIP engine): numpy's legacy RandomState stream
is stable across numpy versions, so a fixture need not carry a 3 MB state_dict — only the seed.
Xavier-uniform weights (bound sqrt(6/(fan_in+fan_out)), dnn.py:47), zero hidden biases (dnn.py:51-52),
last-layer bias U(+-1/sqrt(fan_in)) (nn.Linear's default, which dnn.py:33 leaves in place)."""
import numpy as np


def xavier_params(layers, seed):
    """-> list of float32 arrays [W_0, b_0, W_1, b_1, ...], W_l of shape (out_l, in_l)."""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(len(layers) - 1):
        fi, fo = layers[i], layers[i + 1]
        a = np.sqrt(6.0 / (fi + fo))
        out.append(rng.uniform(-a, a, size=(fo, fi)).astype(np.float32))
        if i < len(layers) - 2:
            out.append(np.zeros(fo, np.float32))
        else:
            b = 1.0 / np.sqrt(fi)
            out.append(rng.uniform(-b, b, size=(fo,)).astype(np.float32))
    return out


def state_dict_of(params):
    """reference state_dict naming: layers.layer_{i}.weight / .bias (dnn.py:32-35)."""
    sd = {}
    for i in range(len(params) // 2):
        sd[f"layers.layer_{i}.weight"] = params[2 * i]
        sd[f"layers.layer_{i}.bias"] = params[2 * i + 1]
    return sd
