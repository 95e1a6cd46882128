"""trainer — the `pinn` training harness of train.py:46-200 / train_newmethod.py:46-209 on the
HIP engine: same constructor arguments, same loss arithmetic, same Adam+StepLR then one
LBFGS.step(closure) schedule, same log.txt / model_{iter}.pth artefacts.

One closure evaluation =
    grad <- 0
    fidelity   : pinn_mse_loss_grad       (train.py:131-141)
    residual   : pinn_residual_loss_grad  (train.py:144-154 + loss.backward(), :191)
    all-reduce : [grad | loss sums] over the data-parallel group (parallel.py)
The reference logs every call (train.py:160-173), i.e. one host synchronisation per iteration.
Here the three loss values of a logged iteration are copied into a device-side ring and written
out `log_flush_every` entries at a time (one synchronisation per flush; `log_flush_every=1`
restores line-by-line behaviour): log.txt has the same lines, just written in batches.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from ._lib import PinnError
from .config import PinnConfig, load_config
from .dnn import DNN
from .engine import ACTIVATION_OF_INIT, Engine, NetDesc, ResidualSpec
from .lbfgs import FlatLBFGS
from .parallel import Reducer


def _as_f32(a, device) -> Optional[torch.Tensor]:
    if a is None:
        return None
    t = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a)
    return t.to(device=device, dtype=torch.float32).contiguous()


class HipEvaluator:
    """Local-shard loss sums and gradient through libpinn_hip.so."""

    def __init__(self, cfg_layers, init_type, grad_cols, spec: ResidualSpec, fid_cols: Sequence[int], device,
                 engine: int = 0, precision: int = 0, dropout_rate: float = 0.0):
        act = ACTIVATION_OF_INIT[init_type]
        self.dropout_rate = float(dropout_rate)
        self.eng = Engine(NetDesc.from_layers(cfg_layers, grad_cols, act, engine, precision), device)
        # training-mode dropout (dnn.py:38 with train.py:186): its own engine (generic kernels carry the mask)
        self.eng_drop = Engine(NetDesc.from_layers(cfg_layers, grad_cols, act, 0, 0, self.dropout_rate), device) \
            if self.dropout_rate > 0.0 else None
        self.training = True
        self.spec, self.fid_cols = spec, list(fid_cols)
        self.merge_sets = True      # one launch for both loss terms when the fidelity set is small
        self._cat = None            # [collocation points ; fidelity points], allocated once
        self._cat_src = (None, None)   # the very tensor OBJECTS whose rows _cat currently holds

    MERGE_MAX_FID = 2048   # fidelity points ride through the jet kernel (4x their own work): only when few

    def __call__(self, theta, Xf, Tf, fid_scale, Xr, res_scale, grad, fid_sums, res_sums):
        if self.eng_drop is not None and self.training:
            # the reference runs the network twice per loss_func call (train.py:133,148): two forward passes, two
            # masks — one fresh seed per pass, from torch's CPU generator
            e = self.eng_drop
            if Xf is not None and Xf.shape[0] > 0:
                e.dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
                if Xf is Xr:
                    e.residual_mse_loss_grad(self.spec, res_scale, Tf, self.fid_cols, fid_scale, theta, Xr, grad,
                                             term_sums=res_sums, col_sums=fid_sums)
                    return
                e.mse_loss_grad(theta, Xf, Tf, self.fid_cols, fid_scale, grad, sums=fid_sums)
            else:
                fid_sums.zero_()
            if Xr is not None and Xr.shape[0] > 0:
                e.dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
                e.residual_loss_grad(self.spec, res_scale, theta, Xr, grad, sums=res_sums)
            else:
                res_sums.zero_()
            return
        if (self.merge_sets and Xf is not None and Xr is not None and Xf is not Xr
                and 0 < Xf.shape[0] <= self.MERGE_MAX_FID and Xr.shape[0] > 0):
            # train.py:131-157 in ONE launch: collocation points first, fidelity points after them
            cat = self._merged(Xr, Xf)
            self.eng.residual_mse_split_loss_grad(self.spec, res_scale, Tf, self.fid_cols, fid_scale, theta, cat,
                                                  Xr.shape[0], grad, term_sums=res_sums, col_sums=fid_sums)
            return
        if Xf is not None and Xf is Xr and Xr.shape[0] > 0:
            # one point set for both terms (train_newmethod.py:122-159): one pass, one forward
            self.eng.residual_mse_loss_grad(self.spec, res_scale, Tf, self.fid_cols, fid_scale, theta, Xr, grad,
                                            term_sums=res_sums, col_sums=fid_sums)
            return
        if Xf is not None and Xf.shape[0] > 0:
            self.eng.mse_loss_grad(theta, Xf, Tf, self.fid_cols, fid_scale, grad, sums=fid_sums)
        else:
            fid_sums.zero_()
        if Xr is not None and Xr.shape[0] > 0:
            self.eng.residual_loss_grad(self.spec, res_scale, theta, Xr, grad, sums=res_sums)
        else:
            res_sums.zero_()

    def adam_step(self, theta, grad, m, v, step, lr):
        self.eng.adam_step(theta, grad, m, v, step, lr)

    def adam_iteration(self, theta, Xf, Tf, fid_scale, Xr, res_scale, grad, fid_sums, res_sums, m, v, step, lr,
                       loss_rows=None, losses=None, params_token=None) -> bool:
        """loss_func + backward + Adam.step (train.py:189-193) in two launches where the engine can
        (Engine.loss_grad_adam_step); False — nothing done — otherwise: the caller then runs __call__ + adam_step."""
        if (self.eng_drop is not None and self.training) or Xr is None or Xr.shape[0] == 0:
            return False
        has_fid = Xf is not None and Xf.shape[0] > 0
        if not has_fid:
            if fid_sums.numel():       # (a configuration with fidelity outputs but no fidelity points: classic path)
                return False
            return self.eng.loss_grad_adam_step(self.spec, res_scale, theta, Xr, Xr.shape[0], grad, m, v, step, lr,
                                                term_sums=res_sums, loss_rows=loss_rows, losses=losses, params_token=params_token)
        if Xf is Xr:                                       # train_newmethod.py:122-159: one point set for both terms
            return self.eng.loss_grad_adam_step(self.spec, res_scale, theta, Xr, -1, grad, m, v, step, lr, T=Tf,
                                                out_col=self.fid_cols, col_scale=fid_scale, term_sums=res_sums,
                                                col_sums=fid_sums, loss_rows=loss_rows, losses=losses, params_token=params_token)
        if not (self.merge_sets and Xf.shape[0] <= self.MERGE_MAX_FID):
            return False
        cat = self._merged(Xr, Xf)
        return self.eng.loss_grad_adam_step(self.spec, res_scale, theta, cat, Xr.shape[0], grad, m, v, step, lr, T=Tf,
                                            out_col=self.fid_cols, col_scale=fid_scale, term_sums=res_sums,
                                            col_sums=fid_sums, loss_rows=loss_rows, losses=losses, params_token=params_token)

    def _merged(self, Xr, Xf):
        """[collocation points ; fidelity points] in one matrix, refreshed whenever either source is a different
        tensor OBJECT than the one it was filled from (held here, so its storage cannot be recycled under us): a
        resampled mini-batch is a new tensor every call and is copied in every call.  data_ptr() is no identity —
        the caching allocator hands a freed block to the next same-sized tensor."""
        nr, nf = Xr.shape[0], Xf.shape[0]
        if self._cat is None or self._cat.shape[0] != nr + nf:
            self._cat = torch.empty(nr + nf, Xr.shape[1], dtype=torch.float32, device=Xr.device)
            self._cat_src = (None, None)
        if self._cat_src[0] is not Xr:
            self._cat[:nr].copy_(Xr)
        if self._cat_src[1] is not Xf:
            self._cat[nr:].copy_(Xf)
        self._cat_src = (Xr, Xf)
        return self._cat

    def predict(self, theta, X):
        return self.eng.forward(theta, X)


class PINN:
    """The physics-guided network harness (reference `class pinn`, train.py:46)."""

    def __init__(self, fidelity_input, fidelity_true, residual_input, config, residual: Optional[str] = None,
                 device="cuda", log_dir: Optional[str] = None, log_every: int = 1, checkpoint_every: int = 1000,
                 reducer: Optional[Reducer] = None, evaluator: Optional[Callable] = None,
                 dnn: Optional[DNN] = None, engine: int = 0, mat_dump_iter: Optional[int] = None,
                 mat_dump_path: str = "data_at50k.mat", residual_batch: Optional[int] = None, seed: int = 1234,
                 log_flush_every: int = 100, lbfgs_impl: str = "flat", precision: int = 0, fold_adam: bool = True):
        cfg = config if isinstance(config, PinnConfig) else load_config(config)
        self.config, self.device = cfg, torch.device(device)
        self.reducer = reducer or Reducer()
        self.fold_adam, self._adam_folded, self._folded_iters, self._run_losses = bool(fold_adam), False, 0, None
        self._fold_refused = None   # key of the (evaluator, point sets) whose folded request the engine has refused
        self.layers = cfg.layers                                           # train.py:52-56
        self.dnn = dnn if dnn is not None else DNN(cfg.layers, cfg.dropout_rate, cfg.init_type)
        self.dnn.to(self.device)
        self.theta = self.dnn.flat_params()
        self.reducer.broadcast_(self.theta)                                # replicas start identical
        P = self.theta.numel()

        residual = residual or cfg.default_residual()
        self.spec = ResidualSpec.from_names(residual, cfg.residual_inputs, cfg.grad_cols, cfg.residual_outputs)
        # i-th fidelity output is compared with output column i (train.py:137-138, train_newmethod.py:129-131)
        self.fid_cols = list(range(len(cfg.fidelity_outputs)))
        if cfg.variant == "newmethod":
            fid_w = [1.0] * len(self.fid_cols)                             # F.mse_loss, unweighted sum
        else:
            fid_w = [cfg.output_weight(k) for k in cfg.fidelity_outputs]   # train.py:94-95,140
        self.weight_fidelity, self.weight_residual = cfg.weight_fid, cfg.weight_res

        Xf, Tf, Xr = (_as_f32(a, self.device) for a in (fidelity_input, fidelity_true, residual_input))
        if Xf is not None and Xr is not None and (fidelity_input is residual_input or
                                                  (cfg.variant == "newmethod" and Xf.shape == Xr.shape and torch.equal(Xf, Xr))):
            Xf = Xr            # the same point set: the evaluator fuses both loss terms into one pass
        self.n_fid = 0 if Xf is None else Xf.shape[0]
        self.n_res = 0 if Xr is None else Xr.shape[0]
        self.Xr = self.reducer.shard(Xr)
        self.Xf = self.Xr if Xf is Xr else self.reducer.shard(Xf)
        self.Tf = self.reducer.shard(Tf)
        nf, nt = len(self.fid_cols), self.spec.n_terms
        dev = self.device
        self._fid_unit = torch.tensor(fid_w, dtype=torch.float32, device=dev) / max(self.n_fid, 1)
        if residual == "continuity_only":
            xcol = cfg.grad_cols[self.spec.dir_of[0]]
            cnt = (self.Xr[:, xcol] < self.spec.threshold).sum().to(torch.float32).reshape(1)
            self.reducer.allreduce_sum_(cnt)                               # global count of x < 25.5
            self._res_unit = torch.cat([torch.tensor([1.0 / self.n_res], device=dev), 1.0 / cnt,
                                        torch.zeros(1, device=dev)])
        else:
            self._res_unit = torch.full((nt,), 1.0 / max(self.n_res, 1), dtype=torch.float32, device=dev)
        self._fid_scale = (self.weight_fidelity * self._fid_unit).contiguous()
        self._res_scale = (self.weight_residual * self._res_unit).contiguous()
        self.buf = torch.zeros(P + nf + nt, dtype=torch.float32, device=dev)   # ONE all-reduce per closure
        self._loss_mat = None      # (3, nf + nt): [fidelity, residual, total] = _loss_mat @ [fid sums | res sums]
        self.grad = self.buf[:P]
        self._fid_sums, self._res_sums = self.buf[P:P + nf], self.buf[P + nf:]
        self.evaluator = evaluator or HipEvaluator(cfg.layers, cfg.init_type, cfg.grad_cols, self.spec,
                                                   self.fid_cols, dev, engine, precision, cfg.dropout_rate)

        # optional resampled collocation mini-batch (the reference is full-batch only):
        # each closure draws `residual_batch` of this rank's points with a device-side generator
        self.residual_batch = residual_batch
        self._gen = None
        if residual_batch is not None:
            if residual == "continuity_only":
                raise PinnError("residual_batch is not supported with continuity_only's data-dependent mean")
            self._gen = torch.Generator(device=dev).manual_seed(seed + self.reducer.rank)
            self._res_unit = torch.full((nt,), 1.0 / (residual_batch * self.reducer.world), dtype=torch.float32, device=dev)
            self._res_scale = (self.weight_residual * self._res_unit).contiguous()
        self.mat_dump_iter, self.mat_dump_path = mat_dump_iter, mat_dump_path
        self.lbfgs_impl = lbfgs_impl     # "flat": lbfgs.FlatLBFGS (batched recursion); "torch": torch.optim.LBFGS
        self.iter = 0                                                      # train.py:73
        self.adam_maxit = cfg.adam["max_it"]
        self.log_dir, self.log_every, self.checkpoint_every = log_dir, max(int(log_every), 1), checkpoint_every
        self._log_fh = None
        self._history: List[tuple] = []
        self._ring = torch.zeros(max(int(log_flush_every), 1), 3, dtype=torch.float32, device=dev)
        self._ring_iters: List[int] = []
        self.last = None
        self.init_optimizers()

    # ---- optimisers (train.py:100-125) ----------------------------------------------------------
    def init_optimizers(self):
        a, lb = self.config.adam, self.config.lbfgs
        self._adam_m = torch.zeros_like(self.theta)
        self._adam_v = torch.zeros_like(self.theta)
        self._adam_step = 0
        self._sched_steps = 0
        self.theta_param = torch.nn.Parameter(self.theta)       # shares storage with every Linear weight
        lbfgs_cls = FlatLBFGS if self.lbfgs_impl == "flat" else torch.optim.LBFGS
        self.optimizer_LBFGS = lbfgs_cls(
            [self.theta_param], lr=lb["learning_rate"], max_iter=lb["max_it"], max_eval=lb.get("max_evaluation"),
            history_size=lb["history_size"], tolerance_grad=lb["tolerance_grad"],
            tolerance_change=lb["tolerance_change"], line_search_fn=lb["line_search_fn"])

    def current_lr(self) -> float:
        """StepLR(step_size, gamma) stepped once per Adam iteration (train.py:109-113,193)."""
        a = self.config.adam
        return a["learning_rate"] * a["scheduler_gamma"] ** (self._sched_steps // a["scheduler_step_size"])

    # ---- loss (train.py:128-181) ----------------------------------------------------------------------
    def loss_func(self, _adam=None) -> torch.Tensor:
        """Total loss (0-dim device tensor); self.grad holds d loss / d theta afterwards.  `_adam` (adam_step only):
        (exp_avg, exp_avg_sq, step, lr) — the evaluator may then fold the Adam update into the pass's last kernel
        (`self._adam_folded` says whether it did)."""
        self.theta = self.dnn.flat_params()
        if self.mat_dump_iter is not None and self.iter == self.mat_dump_iter:
            self.dump_predictions(self.mat_dump_path)                      # train_newmethod.py:141-153
        if hasattr(self.evaluator, "training"):
            self.evaluator.training = self.dnn.training          # dropout follows the module's mode (train.py:186)
        Xr = self.Xr
        if self.residual_batch is not None:
            idx = torch.randint(0, self.Xr.shape[0], (self.residual_batch,), device=self.device, generator=self._gen)
            Xr = self.Xr.index_select(0, idx)
        self._ensure_loss_mat()
        nxt = self.iter + 1
        logged = nxt % self.log_every == 0 or nxt % 1000 == 0
        # the three losses of a logged iteration are computed straight into the device-side ring
        vec = self._ring[len(self._ring_iters)] if logged else self._loss_vec
        self._adam_folded = _adam is not None and self.evaluator.adam_iteration(
            self.theta, self.Xf, self.Tf, self._fid_scale, Xr, self._res_scale, self.grad, self._fid_sums, self._res_sums,
            *_adam, loss_rows=self._loss_mat, losses=vec, params_token=self.dnn.write_token(self.theta))
        if not self._adam_folded:
            self.buf.zero_()
            self.evaluator(self.theta, self.Xf, self.Tf, self._fid_scale, Xr, self._res_scale, self.grad,
                           self._fid_sums, self._res_sums)
            self.reducer.allreduce_sum_(self.buf)
            torch.mv(self._loss_mat, self.buf[self.theta.numel():], out=vec)                  # one small mat-vec: the three losses
        self.iter += 1                                                                        # train.py:160
        fidelity_loss, residual_loss, loss = vec[0], vec[1], vec[2]
        self.last = (fidelity_loss, residual_loss, loss)
        if logged:
            self._ring_iters.append(self.iter)
            if len(self._ring_iters) == self._ring.shape[0]:
                self.flush_log()
        if self._checkpoint_due(self.iter):
            self.save_checkpoint(f"model_{self.iter}.pth")                                    # train.py:175-179
        return loss

    def _ensure_loss_mat(self):
        if self._loss_mat is None:
            nf = self._fid_sums.numel()
            m = torch.zeros(3, self.buf.numel() - self.theta.numel(), dtype=torch.float32, device=self.device)
            m[0, :nf] = self._fid_unit
            m[1, nf:] = self._res_unit
            m[2] = self.weight_fidelity * m[0] + self.weight_residual * m[1]                  # train.py:157
            self._loss_mat = m.contiguous()
            self._loss_vec = torch.zeros(3, dtype=torch.float32, device=self.device)

    def _checkpoint_due(self, it: int) -> bool:
        if not self.checkpoint_every:
            return False
        every = self.checkpoint_every
        if self.config.variant == "newmethod" and self.checkpoint_every == 1000:
            every = 10000 if it <= 45000 else 1000                                            # train_newmethod.py:181-188
        return it % every == 0

    @property
    def history(self) -> List[tuple]:
        """(iteration, fidelity, residual, total) of every logged iteration so far."""
        self.flush_log()
        return self._history

    @history.setter
    def history(self, value):
        self._ring_iters = []
        self._history = list(value)

    def flush_log(self):
        """Bring the pending ring entries to the host (ONE synchronisation) and write them out."""
        if not self._ring_iters:
            return
        rows = self._ring[:len(self._ring_iters)].cpu().tolist()
        iters, self._ring_iters = self._ring_iters, []
        for it, (fid, res, tot) in zip(iters, rows):
            self._log(it, fid, res, tot)
        if self._log_fh is not None:
            self._log_fh.flush()

    def _log(self, it: int, fid: float, res: float, tot: float):
        self._history.append((it, fid, res, tot))
        if it % 1000 == 0 and self.reducer.rank == 0:
            print(f"Epoch {it}, Fidelity Loss: {fid:.5e}, Residual Loss: {res:.5e}, Total Loss: {tot:.5e}")
        if self.log_dir is None or self.reducer.rank != 0:
            return
        if self._log_fh is None:
            os.makedirs(self.log_dir, exist_ok=True)
            path = os.path.join(self.log_dir, "log.txt")
            new = not os.path.exists(path) or os.stat(path).st_size == 0
            self._log_fh = open(path, "a")
            if new:
                self._log_fh.write("Epoch, Fidelity Loss, Residual Loss, Total Loss\n")       # train.py:167
        self._log_fh.write(f"{it}, {fid:.5e}, {res:.5e}, {tot:.5e}\n")                        # train.py:170

    def dump_predictions(self, path: str):
        """savemat of pred_<key> (N,1) float32 for every network output on ALL residual points — the
        file format of the reference's data_at50k.mat (train_newmethod.py:141-153).  Under data
        parallelism every rank predicts its shard, the shards are gathered in rank order (= the
        original row order, parallel.shard_bounds) and rank 0 alone writes the file."""
        Y = self.reducer.gather_rows(self.predict(self.Xr).detach(), self.n_res)
        if self.reducer.rank != 0:
            return
        from scipy.io import savemat
        Y = Y.cpu().numpy().astype(np.float32)
        names = self.config.residual_outputs
        savemat(path, {f"pred_{k}": Y[:, i:i + 1] for i, k in enumerate(names)})
        print(f"Data saved to {path} after {self.iter} iterations.")

    def save_checkpoint(self, name: str):
        self.flush_log()
        if self.log_dir is None or self.reducer.rank != 0:
            return
        os.makedirs(self.log_dir, exist_ok=True)
        torch.save(self.dnn, os.path.join(self.log_dir, name))                    # whole module, as train.py:179
        torch.save(self.dnn.state_dict(), os.path.join(self.log_dir, name.replace(".pth", ".state.pth")))

    # ---- training (train.py:185-200) ------------------------------------------------------------------
    def adam_step(self):
        """zero_grad / loss_func / backward / Adam.step / StepLR.step (train.py:189-193)."""
        # One process, no checkpoint inside this iteration (the reference saves the PRE-update weights from inside
        # loss_func, train.py:175-179): the evaluator may fold the update into the pass's last kernel.
        fold = (self._may_fold() and not self._checkpoint_due(self.iter + 1))
        loss = self.loss_func((self._adam_m, self._adam_v, self._adam_step + 1, self.current_lr()) if fold else None)
        self._adam_step += 1
        if fold and not self._adam_folded:
            self._fold_refused = self._fold_key()      # remembered: no second refused C call per iteration from now on
        if not self._adam_folded:
            self.evaluator.adam_step(self.theta, self.grad, self._adam_m, self._adam_v, self._adam_step,
                                     self.current_lr())
        else:
            self._folded_iters += 1
        self._sched_steps += 1
        return loss

    MAX_RUN = 256      # iterations enqueued by one call (pinn_adam_loop)

    def _fold_key(self):
        return (id(self.evaluator), id(self.Xr), id(self.Xf), self.residual_batch, self.dnn.training)   # (dropout folds in eval mode only)

    def _may_fold(self) -> bool:
        """The folded iteration is worth asking for: switched on, one process, an evaluator that has it, and the engine
        has not already refused this very request (wide / generic engine, a large split request: the refusal is remembered
        per evaluator and point sets instead of being re-discovered by two refused C calls every iteration)."""
        return (self.fold_adam and not self.reducer.active and hasattr(self.evaluator, "adam_iteration")
                and self._fold_refused != self._fold_key())

    def _foldable_run(self, n: int) -> int:
        """How many of the next n Adam iterations can be enqueued by ONE call: full batch, one process, nothing the
        host must do in between (a checkpoint is saved from inside loss_func with pre-update weights, train.py:175-179;
        the prediction dump of train_newmethod.py:141-153 happens at the start of its iteration)."""
        if not (self._may_fold() and self.residual_batch is None):
            return 0
        k = min(n, self.MAX_RUN)
        for i in range(1, k + 1):
            if self._checkpoint_due(self.iter + i):
                k = i - 1
                break
        if self.mat_dump_iter is not None and self.iter <= self.mat_dump_iter < self.iter + k:
            k = self.mat_dump_iter - self.iter
        free, logged = self._ring.shape[0] - len(self._ring_iters), 0
        for i in range(1, k + 1):
            it = self.iter + i
            if it % self.log_every == 0 or it % 1000 == 0:
                logged += 1
                if logged > free:
                    k = i - 1
                    break
        return k

    def train_adam(self, n: int):
        """n Adam iterations (train.py:188-193).  Runs of iterations that need nothing from the host are enqueued by one
        call (two launches per iteration, no Python in between); the rest go through adam_step()."""
        while n > 0:
            k = self._foldable_run(n)
            if k > 1:
                if self._adam_run(k):
                    n -= k
                    continue
                self._fold_refused = self._fold_key()
            self.adam_step()
            n -= 1

    def _adam_run(self, k: int) -> bool:
        self.theta = self.dnn.flat_params()
        self._ensure_loss_mat()
        if hasattr(self.evaluator, "training"):
            self.evaluator.training = self.dnn.training
        a = self.config.adam
        lrs = [a["learning_rate"] * a["scheduler_gamma"] ** ((self._sched_steps + i) // a["scheduler_step_size"]) for i in range(k)]
        if self._run_losses is None or self._run_losses.shape[0] < k:
            self._run_losses = torch.zeros(max(k, self.MAX_RUN), 3, dtype=torch.float32, device=self.device)
        out = self._run_losses[:k]
        if not self.evaluator.adam_iteration(self.theta, self.Xf, self.Tf, self._fid_scale, self.Xr, self._res_scale, self.grad,
                                             self._fid_sums, self._res_sums, self._adam_m, self._adam_v, self._adam_step + 1,
                                             lrs, loss_rows=self._loss_mat, losses=out, params_token=self.dnn.write_token(self.theta)):
            return False
        if self.log_every == 1:                      # every iteration logged: one copy into the ring
            r0 = len(self._ring_iters)
            self._ring[r0:r0 + k].copy_(out)
            self._ring_iters.extend(range(self.iter + 1, self.iter + k + 1))
        else:
            for i in range(k):
                it = self.iter + 1 + i
                if it % self.log_every == 0 or it % 1000 == 0:
                    self._ring[len(self._ring_iters)].copy_(out[i])
                    self._ring_iters.append(it)
        self.iter += k
        self._adam_step += k
        self._sched_steps += k
        self._folded_iters += k
        self._adam_folded = True
        last = out[k - 1].clone()
        self.last = (last[0], last[1], last[2])
        if len(self._ring_iters) == self._ring.shape[0]:
            self.flush_log()
        return True

    def closure(self):
        """train.py:195-199"""
        loss = self.loss_func()
        self.theta_param.grad = self.grad.clone()
        return loss

    def train(self):
        self.dnn.train()
        self.train_adam(self.adam_maxit)
        if self.config.lbfgs["max_it"] > 0:
            self.optimizer_LBFGS.step(self.closure)                                # ONE step, train.py:200
        self.flush_log()

    def predict(self, inputs) -> torch.Tensor:
        """Forward on a grid (test.py:76): (N, d_in) -> (N, d_out)."""
        X = _as_f32(inputs, self.device)
        return self.evaluator.predict(self.dnn.flat_params(), X)


pinn = PINN   # the reference's class name (train.py:46)
