"""dnn — drop-in for the reference's dnn.py (class DNN), running on the HIP engine.

Same constructor, same state_dict keys (layers.layer_{i}.weight / .bias), same
initialisation (dnn.py:42-52), nn.Module semantics.  Differences, all deliberate:
  * the Linear weights are views into ONE flat fp32 buffer [W0,b0,W1,b1,...] — the layout
    the C-ABI takes — so optimisers updating the Parameters in place update it too;
  * forward() runs on libpinn_hip.so and REQUIRES a GPU tensor: there is no CPU path;
  * Dropout(p>0) in training mode (dnn.py:38, train.py:186) uses the engine's counter-based mask (one fresh
    seed per forward call, drawn from torch's CPU generator; include/pinn_hip.h pinn_desc.dropout_p): same
    distribution as nn.Dropout, not torch's random stream.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from ._lib import PinnError
from .engine import ACTIVATION_OF_INIT


def _check_init_type(init_type: str):
    if init_type not in ACTIVATION_OF_INIT:
        # same exception type and text as dnn.py:23 / :49
        raise ValueError(f"Invalid init_type: {init_type}. Use 'kaiming' or 'xavier'.")


def init_flat_params(layers: Sequence[int], init_type: str = "xavier", generator=None) -> torch.Tensor:
    """Flat [W0,b0,...] fp32 vector drawn from the distributions of dnn.py:42-52
    (Xavier/Kaiming-uniform weights; zero bias except the last layer's nn.Linear default)."""
    _check_init_type(init_type)
    parts = []
    n_lin = len(layers) - 1
    for i in range(n_lin):
        fan_in, fan_out = layers[i], layers[i + 1]
        if init_type == "xavier":
            bound = math.sqrt(6.0 / (fan_in + fan_out))
        else:
            # kaiming_uniform_(w, nonlinearity='leaky_relu') leaves a at its default 0: gain sqrt(2/(1+0^2))
            # (dnn.py:45 — NOT the 0.01 slope of the LeakyReLU it then applies)
            bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
        parts.append(torch.empty(fan_out * fan_in).uniform_(-bound, bound, generator=generator))
        if i < n_lin - 1:
            parts.append(torch.zeros(fan_out))
        else:
            b = 1.0 / math.sqrt(fan_in)
            parts.append(torch.empty(fan_out).uniform_(-b, b, generator=generator))
    return torch.cat(parts)


class DNN(nn.Module):
    """The deep neural network (reference dnn.py:5-55)."""

    def __init__(self, layers, dropout_rate, init_type):
        super().__init__()
        _check_init_type(init_type)
        self.activation = nn.Tanh() if init_type == "xavier" else nn.LeakyReLU(negative_slope=0.01)
        self.layer_sizes: List[int] = [int(v) for v in layers]
        self.dropout_rate = float(dropout_rate)
        self.init_type = init_type
        mods = []
        n = len(self.layer_sizes)
        for i in range(n - 1):
            lin = nn.Linear(self.layer_sizes[i], self.layer_sizes[i + 1])
            if init_type == "kaiming":
                nn.init.kaiming_uniform_(lin.weight, nonlinearity="leaky_relu")
            else:
                nn.init.xavier_uniform_(lin.weight)
            if i < n - 2:
                nn.init.zeros_(lin.bias)
            mods.append((f"layer_{i}", lin))
            if i < n - 2:
                mods.append((f"activation_{i}", self.activation))
                mods.append((f"dropout_{i}", nn.Dropout(self.dropout_rate)))
        self.layers = nn.Sequential(OrderedDict(mods))
        self._flat: Optional[torch.Tensor] = None
        self._grad_cols_override = None
        self._flatten()

    # ---- flat parameter storage -----------------------------------------------------------
    def _linears(self):
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def _ordered_params(self):
        # cached: the trainer asks for this list (write_token, flat_params) on every iteration of a host-bound loop
        ps = self.__dict__.get("_plist")
        if ps is None:
            ps = []
            for lin in self._linears():
                ps += [lin.weight, lin.bias]
            self.__dict__["_plist"] = ps
        return ps

    def _flatten(self):
        self.__dict__.pop("_plist", None)      # the slow path re-reads the module tree (a layer replaced since the last time)
        ps = self._ordered_params()
        flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in ps])
        off = 0
        for p in ps:
            n = p.numel()
            p.data = flat[off:off + n].view_as(p)
            off += n
        self._flat = flat

    def _aliased(self) -> bool:
        flat = self._flat
        if flat is None:
            return False
        off = 0
        for p in self._ordered_params():
            if p.device != flat.device or p.dtype != torch.float32 or \
               p.data_ptr() != flat.data_ptr() + 4 * off or not p.is_contiguous():
                return False
            off += p.numel()
        return True

    def flat_params(self) -> torch.Tensor:
        """The (P,) fp32 buffer every Linear parameter aliases; rebuilt after .to()/.cuda()."""
        if not self._aliased():
            self._flatten()
        return self._flat

    def write_token(self, flat=None):
        """Changes whenever a Parameter of this module (or the flat buffer) is written in place through torch
        (optimizer.step, load_state_dict, clipping, p.mul_()): the version counters of every alias family of the flat
        buffer.  `p.data = flat[...]` gives each Parameter a counter of its own, so the flat buffer's alone would miss
        those writes (Engine.loss_grad_adam_step's packed-weights token)."""
        flat = self._flat if flat is None else flat     # (the caller's flat_params() result: already validated)
        if flat is None:
            flat = self.flat_params()
        return (id(flat), flat._version) + tuple([p._version for p in self._ordered_params()])

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.pop("_plist", None)
        if "layer_sizes" not in self.__dict__:          # a pickle written by the reference's class
            lins = self._linears()
            self.layer_sizes = [lins[0].in_features] + [l.out_features for l in lins]
            self.init_type = "xavier" if isinstance(self.activation, nn.Tanh) else "kaiming"
            drops = [m for m in self.layers if isinstance(m, nn.Dropout)]
            self.dropout_rate = float(drops[0].p) if drops else 0.0
            self._grad_cols_override = None
        self._flat = None

    def set_grad_columns(self, cols: Optional[Sequence[int]]):
        """Name the differentiated input columns explicitly instead of reading them off the
        torch.cat graph (train.py:87-88, 148)."""
        self._grad_cols_override = None if cols is None else tuple(int(c) for c in cols)

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, x):
        if not x.is_cuda:
            raise PinnError("DNN.forward needs a GPU tensor: pinn_depthestimation_amd has no CPU path "
                            "(move the model and its inputs to cuda)")
        from .autograd import dnn_forward
        drop = (0.0, 0)
        if self.training and self.dropout_rate > 0.0:       # a new mask per forward pass, as nn.Dropout draws one
            drop = (self.dropout_rate, int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        return dnn_forward(self, x, drop)
