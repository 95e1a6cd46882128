"""config — reads the reference's JSON configs verbatim (config_CMB.json, config_CMB_h.json and
the older-schema config.json / config_txyz.json) and exposes the few facts the hot path needs.

Reference behaviour kept: sections and key names; `requires_grad` is a LIST OF
STRINGS tested with `"true" in info["requires_grad"]` (train.py:87); float-typed iteration
counts (config.json:17 `5.00e4`, config_CMB.json:21 `6.25e4`) are coerced to int for loops.
Defaults for keys the old schema lacks: dropout_rate 0.0, init_type "xavier"
(train.py:59,62 would raise KeyError there).  Unknown extra keys (e.g. "engine") are ignored.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


def _names(section) -> List[str]:
    """A variable list may be a JSON list (config_CMB.json:42) or a dict keyed by name
    (config_CMB.json:48-51, config.json:40-45)."""
    if section is None:
        return []
    return list(section.keys()) if isinstance(section, dict) else list(section)


def _grad_cols(inputs) -> Tuple[int, ...]:
    if not isinstance(inputs, dict):
        return ()
    cols = []
    for i, (_, info) in enumerate(inputs.items()):
        if isinstance(info, dict) and "true" in info.get("requires_grad", []):   # train.py:87
            cols.append(i)
    return tuple(cols)


@dataclass
class PinnConfig:
    raw: dict
    layers: List[int]
    dropout_rate: float
    init_type: str
    adam: Dict[str, float]
    lbfgs: Dict[str, object]
    loss_weights: Dict[str, float]
    variant: str                           # "train" (train.py) or "newmethod" (train_newmethod.py)
    fidelity_inputs: List[str] = field(default_factory=list)
    fidelity_outputs: List[str] = field(default_factory=list)
    residual_inputs: List[str] = field(default_factory=list)
    residual_outputs: List[str] = field(default_factory=list)
    grad_cols: Tuple[int, ...] = ()
    trues: List[str] = field(default_factory=list)
    unknowns: List[str] = field(default_factory=list)

    @property
    def weight_fid(self) -> float:
        return float(self.loss_weights.get("weight_fid_loss", 1))

    @property
    def weight_res(self) -> float:
        return float(self.loss_weights.get("weight_res_loss", 1))

    def output_weight(self, key: str) -> float:
        """config.loss.weight_<key>_loss (train.py:94-95); 1 when the old schema omits it."""
        return float(self.loss_weights.get(f"weight_{key}_loss", 1))

    def default_residual(self) -> str:
        """The reference hard-codes the residual by import (train.py:17, train_newmethod.py:18);
        pick the one whose argument names the config's variables satisfy."""
        if self.variant == "newmethod":
            return "continuity_only"
        if "t" in self.residual_inputs:
            return "Navier_Stokes"
        if "eta_mean" in self.residual_outputs:
            return "physics_equation"
        return "continuity_ftemp"


def load_config(src) -> PinnConfig:
    raw = src if isinstance(src, dict) else json.load(open(src, "r"))
    L = raw["layers"]
    layers = [int(L["input_features"])] + [int(L["hidden_width"])] * int(L["hidden_layers"]) + \
        [int(L["output_features"])]                                            # train.py:52-56
    adam = dict(raw.get("adam_optimizer", {}))
    adam["max_it"] = int(adam.get("max_it", 0))
    adam.setdefault("learning_rate", 1e-4)
    adam["scheduler_step_size"] = int(adam.get("scheduler_step_size", 10000))
    adam.setdefault("scheduler_gamma", 0.8)
    lb = dict(raw.get("lbfgs_optimizer", {}))
    lb["max_it"] = int(lb.get("max_it", 0))
    if "max_evaluation" in lb and lb["max_evaluation"] is not None:
        lb["max_evaluation"] = int(lb["max_evaluation"])
    lb.setdefault("learning_rate", 1)
    lb.setdefault("history_size", 100)
    lb.setdefault("tolerance_grad", 1e-5)
    lb.setdefault("tolerance_change", 1e-7)
    lb.setdefault("line_search_fn", "strong_wolfe")
    cfg = PinnConfig(raw=raw, layers=layers, dropout_rate=float(L.get("dropout_rate", 0.0)),
                     init_type=L.get("init_type", "xavier"), adam=adam, lbfgs=lb,
                     loss_weights=dict(raw.get("loss", {})), variant="train")
    if "data" in raw and "data_residual" not in raw:          # config_CMB_h.json:33-41
        d = raw["data"]
        cfg.variant = "newmethod"
        cfg.residual_inputs = _names(d["inputs"])
        cfg.grad_cols = _grad_cols(d["inputs"])
        cfg.trues, cfg.unknowns = _names(d.get("trues")), _names(d.get("unknowns"))
        cfg.fidelity_inputs = list(cfg.residual_inputs)
        cfg.fidelity_outputs = list(cfg.trues)
        cfg.residual_outputs = cfg.trues + cfg.unknowns          # train_newmethod.py:136-139
    else:
        df, dr = raw.get("data_fidelity", {}), raw.get("data_residual", {})
        cfg.fidelity_inputs = _names(df.get("inputs"))
        cfg.fidelity_outputs = _names(df.get("outputs"))
        cfg.residual_inputs = _names(dr.get("inputs"))
        cfg.residual_outputs = _names(dr.get("outputs"))
        cfg.grad_cols = _grad_cols(dr.get("inputs"))
    return cfg
