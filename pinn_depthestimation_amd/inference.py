"""inference — the reference's test.py / test_newmethod.py on the HIP engine: forward on the full
ny x nx grid and the optional physics-only L-BFGS fine-tune.

Reference behaviour kept (test.py:10-106): LBFGS(max_iter=1, max_eval=2, history_size=10) with the
config's lr/tolerances/line search (:44-54); inputs become one (N,1) tensor per config variable
with requires_grad from the config strings (:60-65); predictions are reshaped to (ny, nx) as
`plot_pred_<key>` and inputs denormalised to `plot_input_<key>` (:67-84); when
config['perform_optimization'] is true ONE optimizer_LBFGS.step(closure) runs on the residual
alone and the grid is predicted again (:92-104).  Differences: the model may be given as a DNN,
a whole-module file or a state_dict file; the residual follows the config's variable names
(the reference hard-wires Navier_Stokes at test.py:6 while its config_CMB.json has no `t`).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import operations as op
from . import physics
from .config import PinnConfig, load_config
from .dnn import DNN
from .engine import RESIDUAL_ROLES


class Tester:
    def __init__(self, model, config, device="cuda", residual: Optional[str] = None):
        self.config: PinnConfig = config if isinstance(config, PinnConfig) else load_config(config)
        self.device = torch.device(device)
        self.model = self.load_model(model)
        raw = self.config.raw
        dt = raw.get("data_test", {})
        ins = dt.get("inputs", raw.get("data_residual", {}).get("inputs", raw.get("data", {}).get("inputs", {})))
        self.test_input_vars: Dict[str, dict] = ins if isinstance(ins, dict) else {k: {"requires_grad": []} for k in ins}
        outs = dt.get("outputs", self.config.residual_outputs)
        self.test_output_vars = list(outs.keys()) if isinstance(outs, dict) else list(outs)
        self.nx, self.ny = dt.get("nx"), dt.get("ny")
        self.residual = residual or self.config.default_residual()
        self.init_optimizers()
        self.last_loss = None

    def load_model(self, model) -> DNN:
        if isinstance(model, DNN):
            m = model
        else:
            try:
                obj = torch.load(model, map_location="cpu", weights_only=True)       # a state_dict file
            except Exception:
                obj = torch.load(model, map_location="cpu", weights_only=False)      # whole-module pickle (test.py:37)
            if isinstance(obj, DNN):
                m = obj
            else:
                m = DNN(self.config.layers, self.config.dropout_rate, self.config.init_type)
                m.load_state_dict(obj)
        m.to(self.device)
        m.eval()
        return m

    def init_optimizers(self):
        lb = self.config.lbfgs
        self.optimizer_LBFGS = torch.optim.LBFGS(                                   # test.py:44-54
            self.model.parameters(), lr=lb["learning_rate"], max_iter=1, max_eval=2, history_size=10,
            tolerance_grad=lb["tolerance_grad"], tolerance_change=lb["tolerance_change"],
            line_search_fn=lb["line_search_fn"])

    def _residual_loss(self):
        _, out_roles, dir_roles = RESIDUAL_ROLES[self.residual]
        fn = getattr(physics, self.residual)
        return fn(*[getattr(self, k) for k in dir_roles], *[getattr(self, k) for k in out_roles])

    def test(self, test_input_data, input_min_max: Optional[dict] = None, perform_optimization: Optional[bool] = None):
        data = torch.as_tensor(np.asarray(test_input_data)).float().to(self.device)
        cols = []
        for i, (key, info) in enumerate(self.test_input_vars.items()):
            t = data[:, i:i + 1].clone().detach()
            if "true" in info.get("requires_grad", []):
                t = t.requires_grad_()
            setattr(self, key, t)
            cols.append(t)
            if self.nx and self.ny and t.numel() == self.nx * self.ny:
                grid = t.detach().cpu().numpy().reshape(self.ny, self.nx)
                if input_min_max is not None and key in input_min_max:
                    grid = op.denormalize(grid, input_min_max[key][0], input_min_max[key][1])
                setattr(self, f"plot_input_{key}", grid)
        pred = self.model(torch.cat(cols, dim=-1))
        self._publish(pred)
        if perform_optimization is None:
            perform_optimization = bool(self.config.raw.get("perform_optimization", False))
        if perform_optimization:
            def closure():                                                           # test.py:94-99
                self.optimizer_LBFGS.zero_grad()
                loss = self._residual_loss()
                if loss.requires_grad:
                    loss.backward()
                self.last_loss = loss.detach()
                return loss
            self.optimizer_LBFGS.step(closure)
            with torch.no_grad():
                pred = self.model(torch.cat(cols, dim=-1))
            self._publish(pred)
        return pred.detach().cpu().numpy()

    def _publish(self, pred):
        for i, key in enumerate(self.test_output_vars):
            t = pred[:, i:i + 1]
            setattr(self, key, t)
            if self.nx and self.ny and t.numel() == self.nx * self.ny:
                setattr(self, f"plot_pred_{key}", t.detach().cpu().numpy().reshape(self.ny, self.nx))
