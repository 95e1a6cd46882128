"""operations — drop-in for the reference's operations.py (min-max scaling helpers,
operations.py:4-30).  Host NumPy; not on the hot path."""
import numpy as np


def normalize(data, data_min, data_max):
    """Map [data_min, data_max] onto [-1, 1]; a degenerate range maps to zeros (operations.py:4-7)."""
    span = data_max - data_min
    if span == 0:
        return np.zeros_like(data)
    return 2 * (data - data_min) / span - 1


def denormalize(data, data_min, data_max):
    """Inverse of normalize (operations.py:10-13)."""
    span = data_max - data_min
    if span == 0:
        return np.zeros_like(data_min)
    return (data + 1) / 2 * span + data_min


def get_min_max(data, key=None, config=None):
    """operations.py:16-30 defines get_min_max(data, key, config) while train.py:228 and
    test.py:161 call get_min_max(data, config).  Both forms are accepted:
      get_min_max(data, key, config) -> {key: (min, max)}
      get_min_max(data, config)      -> {k: (min, max) for every k in data}
    x / y bounds come from config['data_test'], everything else from the data."""
    if config is None and isinstance(key, dict):
        config, key = key, None
    keys = [key] if key is not None else list(data)
    out = {}
    for k in keys:
        if k in ("x", "y"):
            out[k] = (config["data_test"][f"{k}_min"], config["data_test"][f"{k}_max"])
        else:
            out[k] = (np.nanmin(data[k]), np.nanmax(data[k]))
    return out
