"""Engine — thin host wrapper that hands torch device tensors to the C-ABI.

PyTorch is plumbing here: device memory, the current HIP stream, and (in
parallel.py) torch.distributed.  All arithmetic of the hot path runs in
libpinn_hip.so.  Nothing in this file synchronises the device.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (ACT_LEAKY_RELU, ACT_TANH, ENGINE_AUTO, ENGINE_FUSED, ENGINE_FUSED_BATCH, ENGINE_FUSED_COOP, ENGINE_FUSED_TILE, ENGINE_GENERIC,
                   ENGINE_WIDE, RES_CONTINUITY_FTEMP,
                   RES_CONTINUITY_ONLY, RES_NAVIER_STOKES, RES_PHYSICS_EQUATION, RES_TERMS, PinnDesc, PinnError,
                   PinnResidualSpec, check)

ACTIVATION_OF_INIT = {"xavier": ACT_TANH, "kaiming": ACT_LEAKY_RELU}  # dnn.py:18-21

# residual name -> (id, output role names, direction role names)
RESIDUAL_ROLES = {
    "Navier_Stokes": (RES_NAVIER_STOKES, ("h", "z", "u", "v"), ("t", "x", "y")),          # physics.py:50
    "physics_equation": (RES_PHYSICS_EQUATION, ("h", "U", "V", "eta_mean", "Hrms", "k"), ("x", "y")),  # :91
    "continuity_ftemp": (RES_CONTINUITY_FTEMP, ("h", "U", "V"), ("x", "y")),               # physics.py:37
    "continuity_only": (RES_CONTINUITY_ONLY, ("h", "U", "V"), ("x", "y")),                 # physics.py:18
}


@dataclass(frozen=True)
class NetDesc:
    """Network geometry: layers = [d_in] + [width]*n_hidden + [d_out] (train.py:56)."""
    d_in: int
    d_out: int
    n_hidden: int
    width: int
    grad_cols: Tuple[int, ...] = ()      # X columns whose inputs have requires_grad "true" (train.py:87)
    activation: int = ACT_TANH
    engine: int = ENGINE_AUTO
    precision: int = 0                   # _lib.PREC_F32 / PREC_BF16 (bf16 MFMA operands, wide engine only)
    dropout_p: float = 0.0               # nn.Dropout rate applied after every hidden activation (training mode)

    @property
    def k(self) -> int:
        return len(self.grad_cols)

    @property
    def layers(self) -> List[int]:
        return [self.d_in] + [self.width] * self.n_hidden + [self.d_out]

    @property
    def n_params(self) -> int:
        ls = self.layers
        return sum(ls[i] * ls[i + 1] + ls[i + 1] for i in range(len(ls) - 1))

    def with_(self, **kw) -> "NetDesc":
        d = dict(d_in=self.d_in, d_out=self.d_out, n_hidden=self.n_hidden, width=self.width,
                 grad_cols=self.grad_cols, activation=self.activation, engine=self.engine,
                 precision=self.precision, dropout_p=self.dropout_p)
        d.update(kw)
        return NetDesc(**d)

    def c_struct(self) -> PinnDesc:
        if self.k > _lib.PINN_MAX_DIRS:
            raise PinnError(f"{self.k} differentiated inputs; the engine carries at most {_lib.PINN_MAX_DIRS}")
        d = PinnDesc()
        d.d_in, d.d_out, d.n_hidden, d.width = self.d_in, self.d_out, self.n_hidden, self.width
        d.k = self.k
        for j in range(_lib.PINN_MAX_DIRS):
            d.dir_col[j] = self.grad_cols[j] if j < self.k else -1
        d.activation, d.engine, d.precision = self.activation, self.engine, self.precision
        d.dropout_p, d.dropout_seed = float(self.dropout_p), 0
        return d

    @staticmethod
    def from_layers(layers: Sequence[int], grad_cols=(), activation=ACT_TANH, engine=ENGINE_AUTO, precision=0,
                    dropout_p=0.0) -> "NetDesc":
        if len(layers) < 3 or len(set(layers[1:-1])) != 1:
            raise PinnError(f"layers {list(layers)} are not [d_in] + [width]*n + [d_out] (train.py:56)")
        return NetDesc(layers[0], layers[-1], len(layers) - 2, layers[1], tuple(grad_cols), activation, engine, precision,
                       dropout_p)


@dataclass(frozen=True)
class ResidualSpec:
    """Which output column plays which role in a residual (physics.py signatures)."""
    name: str
    out_col: Tuple[int, ...]
    dir_of: Tuple[int, ...]           # per direction role: index into NetDesc.grad_cols
    threshold: float = 25.5           # physics.py:26
    anchor: float = 0.75              # physics.py:27

    @property
    def residual_id(self) -> int:
        return RESIDUAL_ROLES[self.name][0]

    @property
    def n_terms(self) -> int:
        return RES_TERMS[self.residual_id]

    def c_struct(self) -> PinnResidualSpec:
        s = PinnResidualSpec()
        s.residual_id = self.residual_id
        for r in range(_lib.PINN_MAX_ROLES):
            s.out_col[r] = self.out_col[r] if r < len(self.out_col) else 0
        for d in range(_lib.PINN_MAX_DIRS):
            s.dir_of[d] = self.dir_of[d] if d < len(self.dir_of) else 0
        s.flags = 0
        s.param[0], s.param[1] = self.threshold, self.anchor
        return s

    @staticmethod
    def from_names(name: str, input_names: Sequence[str], grad_cols: Sequence[int],
                   output_names: Sequence[str]) -> "ResidualSpec":
        """Map config variable names (config data_residual.inputs / outputs) onto roles."""
        if name not in RESIDUAL_ROLES:
            raise PinnError(f"unknown residual {name!r}; known: {sorted(RESIDUAL_ROLES)}")
        _, out_roles, dir_roles = RESIDUAL_ROLES[name]
        out_col = []
        for r in out_roles:
            if r not in output_names:
                raise PinnError(f"residual {name} needs output {r!r}; network outputs are {list(output_names)}")
            out_col.append(list(output_names).index(r))
        dir_of = []
        for r in dir_roles:
            if r not in input_names:
                raise PinnError(f"residual {name} needs input {r!r}; network inputs are {list(input_names)}")
            col = list(input_names).index(r)
            if col not in grad_cols:
                raise PinnError(f"input {r!r} must have requires_grad 'true' for residual {name}")
            dir_of.append(list(grad_cols).index(col))
        return ResidualSpec(name, tuple(out_col), tuple(dir_of))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, name: str, shape=None):
    if not t.is_cuda:
        raise PinnError(f"{name} must live on the GPU (got {t.device}); this engine has no CPU path")
    if t.dtype != torch.float32:
        raise PinnError(f"{name} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise PinnError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise PinnError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


class Engine:
    """One network geometry bound to one device; owns the workspace cache."""

    def __init__(self, desc: NetDesc, device: torch.device | str = "cuda"):
        self.lib = _lib.load()
        self.desc = desc
        self.device = torch.device(device)
        self._cdesc: Dict[Tuple[int, int], PinnDesc] = {}
        self._ws: Dict[int, torch.Tensor] = {}
        self._ws_need: Dict[Tuple[int, int], int] = {}
        self._packed_tok = None      # (workspace, params storage, params version, caller's token) after loss_grad_adam_step
        self.dropout_seed = 0        # training-mode dropout: the caller sets a fresh seed per forward pass; the
                                     # reverse sweep of that pass must run under the same one (include/pinn_hip.h)
        cnt = C.c_int64()
        check(self.lib.pinn_param_count(C.byref(desc.c_struct()), C.byref(cnt)), "pinn_param_count")
        self.n_params = cnt.value

    # ---- plumbing -------------------------------------------------------------------
    def _d(self, engine: Optional[int] = None) -> PinnDesc:
        e = self.desc.engine if engine is None else engine
        if e not in self._cdesc:
            self._cdesc[e] = self.desc.with_(engine=e).c_struct()
        d = self._cdesc[e]
        d.dropout_seed = int(self.dropout_seed) & 0xFFFFFFFF
        return d

    def _chk(self, t: torch.Tensor, name: str, shape=None):
        _chk(t, name, shape)
        if t.device.index != self._index():
            raise PinnError(f"{name} lives on {t.device} but this engine is bound to {self.device}")

    def _index(self) -> int:
        """Ordinal of this engine's GPU ("cuda" without an index binds to the device current at first use)."""
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device.index

    def _run(self, what: str, fn, *args):
        """Call one C-ABI entry point on THIS engine's device and on that device's current stream
        (appended as the last argument), whatever device the calling thread has current: launch
        geometry (CU count) and the stream both belong to the device the tensors live on."""
        idx = self._index()
        self._packed_tok = None      # any other call may re-pack the workspace from ITS params argument
        if torch.cuda.current_device() != idx:
            with torch.cuda.device(idx):
                check(fn(*args, C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)), what)
        else:
            check(fn(*args, C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)), what)

    def workspace(self, N: int, engine: Optional[int] = None) -> torch.Tensor:
        e = self.desc.engine if engine is None else engine
        key = (e, N)
        need = self._ws_need.get(key)
        if need is None:
            c_need = C.c_int64()
            idx = self._index()
            if torch.cuda.current_device() != idx:      # the answer depends on the device's CU count
                with torch.cuda.device(idx):
                    check(self.lib.pinn_query_workspace(C.byref(self._d(e)), N, C.byref(c_need)), "pinn_query_workspace")
            else:
                check(self.lib.pinn_query_workspace(C.byref(self._d(e)), N, C.byref(c_need)), "pinn_query_workspace")
            need = self._ws_need[key] = c_need.value
        ws = self._ws.get(e)
        if ws is None or ws.numel() < need:
            self._ws[e] = ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        return ws

    # ---- calls ------------------------------------------------------------------------
    def forward(self, params: torch.Tensor, X: torch.Tensor, engine=None) -> torch.Tensor:
        N = X.shape[0]
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in))
        Y = torch.empty(N, self.desc.d_out, dtype=torch.float32, device=X.device)
        ws = self.workspace(N, engine)
        self._run("pinn_forward", self.lib.pinn_forward, C.byref(self._d(engine)), _ptr(params), _ptr(X), N, _ptr(Y), _ptr(ws),
                  ws.numel())
        return Y

    def forward_jet(self, params: torch.Tensor, X: torch.Tensor, engine=None):
        N = X.shape[0]
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in))
        Y = torch.empty(N, self.desc.d_out, dtype=torch.float32, device=X.device)
        dY = torch.empty(self.desc.k, N, self.desc.d_out, dtype=torch.float32, device=X.device)
        ws = self.workspace(N, engine)
        self._run("pinn_forward_jet", self.lib.pinn_forward_jet, C.byref(self._d(engine)), _ptr(params), _ptr(X), N, _ptr(Y), _ptr(dY),
                                        _ptr(ws), ws.numel())
        return Y, dY

    def jet_backward(self, params, X, gY: Optional[torch.Tensor], gdY: Optional[torch.Tensor],
                     grad: torch.Tensor) -> torch.Tensor:
        N = X.shape[0]
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in))
        self._chk(grad, "grad", (self.n_params,))
        if gY is not None: self._chk(gY, "gY", (N, self.desc.d_out))
        if gdY is not None: self._chk(gdY, "gdY", (self.desc.k, N, self.desc.d_out))
        ws = self.workspace(N, ENGINE_GENERIC)
        self._run("pinn_jet_backward", self.lib.pinn_jet_backward, C.byref(self._d(ENGINE_GENERIC)), _ptr(params), _ptr(X), N, _ptr(gY),
                                         _ptr(gdY), _ptr(grad), _ptr(ws), ws.numel())
        return grad

    def residual_loss(self, spec: ResidualSpec, params, X, engine=None) -> torch.Tensor:
        N = X.shape[0]
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in))
        sums = torch.empty(spec.n_terms, dtype=torch.float32, device=X.device)
        ws = self.workspace(N, engine)
        self._run("pinn_residual_loss", self.lib.pinn_residual_loss, C.byref(self._d(engine)), C.byref(spec.c_struct()), _ptr(params),
                                          _ptr(X), N, _ptr(sums), _ptr(ws), ws.numel())
        return sums

    def residual_loss_grad(self, spec: ResidualSpec, term_scale: torch.Tensor, params, X, grad: torch.Tensor,
                           engine=None, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
        """grad += sum_t term_scale[t] * d(term_sums[t])/d(params); returns term_sums (device)."""
        N = X.shape[0]
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in))
        self._chk(grad, "grad", (self.n_params,)); self._chk(term_scale, "term_scale", (spec.n_terms,))
        if sums is None:
            sums = torch.empty(spec.n_terms, dtype=torch.float32, device=X.device)
        ws = self.workspace(N, engine)
        self._run("pinn_residual_loss_grad", self.lib.pinn_residual_loss_grad, C.byref(self._d(engine)), C.byref(spec.c_struct()),
                                               _ptr(term_scale), _ptr(params), _ptr(X), N, _ptr(sums),
                                               _ptr(grad), _ptr(ws), ws.numel())
        return sums

    def mse_loss_grad(self, params, X, T: torch.Tensor, out_col: Sequence[int],
                      col_scale: Optional[torch.Tensor], grad: Optional[torch.Tensor], engine=None,
                      sums: Optional[torch.Tensor] = None) -> torch.Tensor:
        N, nc = X.shape[0], len(out_col)
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in)); self._chk(T, "T", (N, nc))
        if grad is not None:
            self._chk(grad, "grad", (self.n_params,)); self._chk(col_scale, "col_scale", (nc,))
        if sums is None:
            sums = torch.empty(nc, dtype=torch.float32, device=X.device)
        oc = (C.c_int32 * nc)(*out_col)
        ws = self.workspace(N, engine)
        self._run("pinn_mse_loss_grad", self.lib.pinn_mse_loss_grad, C.byref(self._d(engine)), _ptr(params), _ptr(X), _ptr(T), N, nc, oc,
                                          _ptr(col_scale), _ptr(sums), _ptr(grad), _ptr(ws), ws.numel())
        return sums

    def residual_mse_loss_grad(self, spec: ResidualSpec, term_scale, T: torch.Tensor, out_col: Sequence[int],
                               col_scale, params, X, grad, engine=None, term_sums=None, col_sums=None):
        """One pass over one point set: PDE residual + fidelity columns (train_newmethod.py:122-159)."""
        N, nc = X.shape[0], len(out_col)
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in)); self._chk(T, "T", (N, nc))
        self._chk(grad, "grad", (self.n_params,)); self._chk(term_scale, "term_scale", (spec.n_terms,))
        self._chk(col_scale, "col_scale", (nc,))
        if term_sums is None:
            term_sums = torch.empty(spec.n_terms, dtype=torch.float32, device=X.device)
        if col_sums is None:
            col_sums = torch.empty(nc, dtype=torch.float32, device=X.device)
        oc = (C.c_int32 * nc)(*out_col)
        ws = self.workspace(N, engine)
        self._run("pinn_residual_mse_loss_grad", self.lib.pinn_residual_mse_loss_grad, C.byref(self._d(engine)), C.byref(spec.c_struct()), _ptr(term_scale),
                                                   _ptr(T), nc, oc, _ptr(col_scale), _ptr(params), _ptr(X), N,
                                                   _ptr(term_sums), _ptr(col_sums), _ptr(grad), _ptr(ws), ws.numel())
        return term_sums, col_sums

    def residual_mse_split_loss_grad(self, spec: ResidualSpec, term_scale, T: torch.Tensor, out_col: Sequence[int],
                                     col_scale, params, X, n_res: int, grad, engine=None, term_sums=None,
                                     col_sums=None):
        """train.py:131-157 in one launch: X = [n_res collocation points ; fidelity points], T = the
        fidelity targets (N - n_res, n_cols)."""
        N, nc = X.shape[0], len(out_col)
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in)); self._chk(T, "T", (N - n_res, nc))
        self._chk(grad, "grad", (self.n_params,)); self._chk(term_scale, "term_scale", (spec.n_terms,))
        self._chk(col_scale, "col_scale", (nc,))
        if term_sums is None:
            term_sums = torch.empty(spec.n_terms, dtype=torch.float32, device=X.device)
        if col_sums is None:
            col_sums = torch.empty(nc, dtype=torch.float32, device=X.device)
        oc = (C.c_int32 * nc)(*out_col)
        ws = self.workspace(N, engine)
        self._run("pinn_residual_mse_split_loss_grad", self.lib.pinn_residual_mse_split_loss_grad, 
            C.byref(self._d(engine)), C.byref(spec.c_struct()), _ptr(term_scale), _ptr(T), nc, oc, _ptr(col_scale),
            _ptr(params), _ptr(X), N, int(n_res), _ptr(term_sums), _ptr(col_sums), _ptr(grad), _ptr(ws), ws.numel())
        return term_sums, col_sums

    def loss_grad_adam_step(self, spec: ResidualSpec, term_scale, params, X, n_res: int, grad, m, v, step: int, lr,
                            T: Optional[torch.Tensor] = None, out_col: Sequence[int] = (), col_scale=None,
                            term_sums=None, col_sums=None, beta1=0.9, beta2=0.999, eps=1e-8,
                            loss_rows: Optional[torch.Tensor] = None, losses: Optional[torch.Tensor] = None,
                            params_token=None) -> bool:
        """train.py:189-193 in two launches (pinn_loss_grad_adam_step): loss + gradient at `params`, then ONE kernel
        that finishes sums and gradient, applies Adam to params / m / v and refreshes the packed weights of this N's
        workspace; with `loss_rows` (rows x (len(out_col) + n_terms)) it also writes losses = loss_rows @ [col sums | term sums].
        `lr` a sequence of n learning rates: n consecutive iterations (steps step .. step + n - 1) enqueued by ONE call
        (pinn_adam_loop), iteration i writing losses[i].
        Returns False — nothing launched — when the request is not a one-pass request of the fused engine.
        The packing kernel is skipped when the previous call on this engine was this method and `params` has not
        been written since: same workspace, same storage, same torch version counter of `params` AND the same
        `params_token`.  torch's version counter is per alias family: tensors that share `params`' storage through
        `.data` (dnn.DNN's Linear weights, `p.data = flat[...]`) have counters of their own, so a caller whose buffer
        is aliased that way passes a token that changes whenever any alias is written (DNN.write_token(): the
        Parameters' version counters); writes no counter sees (`p.data.mul_()`) need invalidate_packed()."""
        N, nc = X.shape[0], len(out_col)
        lrs = [float(x) for x in lr] if isinstance(lr, (list, tuple)) else None
        self._chk(params, "params", (self.n_params,)); self._chk(X, "X", (N, self.desc.d_in))
        for t, nme in ((grad, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            self._chk(t, nme, (self.n_params,))
        self._chk(term_scale, "term_scale", (spec.n_terms,)); self._chk(term_sums, "term_sums", (spec.n_terms,))
        if nc:
            self._chk(col_scale, "col_scale", (nc,)); self._chk(col_sums, "col_sums", (nc,))
            if n_res != N:
                self._chk(T, "T", (N - n_res if n_res >= 0 else N, nc))
        n_rows = 0
        if loss_rows is not None:
            n_rows = loss_rows.shape[0]
            self._chk(loss_rows, "loss_rows", (n_rows, nc + spec.n_terms))
            self._chk(losses, "losses", (n_rows,) if lrs is None else (len(lrs), n_rows))
        ws = self.workspace(N)
        tok = self._packed_tok
        packed_valid = (tok is not None and tok[0] is ws and tok[1] == params.data_ptr() and tok[2] == params._version
                        and tok[3] == params_token and tok[4] == (N, int(n_res), nc))     # (the request picks the kernel, and with it the packed layout)
        self._packed_tok = None
        st = _lib.PinnAdamState(_ptr(m), _ptr(v), int(step), 0.0 if lrs is not None else float(lr), float(beta1), float(beta2),
                                float(eps), 1 if packed_valid else 0, n_rows, _ptr(loss_rows), _ptr(losses))
        oc = (C.c_int32 * max(nc, 1))(*out_col)
        idx = self._index()
        head = (C.byref(self._d()), C.byref(spec.c_struct()), _ptr(term_scale), _ptr(T), nc, oc, _ptr(col_scale),
                _ptr(params), _ptr(X), N, int(n_res), _ptr(term_sums), _ptr(col_sums), _ptr(grad), C.byref(st))
        with torch.cuda.device(idx):
            tail = (_ptr(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream(idx).cuda_stream))
            if lrs is None:
                rc = self.lib.pinn_loss_grad_adam_step(*head, *tail)
            else:
                rc = self.lib.pinn_adam_loop(*head, len(lrs), (C.c_double * len(lrs))(*lrs), *tail)
        if rc == _lib.ERR_UNSUPPORTED:
            return False
        check(rc, "pinn_loss_grad_adam_step" if lrs is None else "pinn_adam_loop")
        self._packed_tok = (ws, params.data_ptr(), params._version, params_token, (N, int(n_res), nc))
        return True

    def invalidate_packed(self):
        """Forget the packed copy of the parameters left in the workspace by loss_grad_adam_step: the next call re-packs.
        For writers torch's version counters cannot see (a `.data` view written in place, a raw pointer)."""
        self._packed_tok = None

    def adam_step(self, params, grad, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
        for t, nme in ((params, "params"), (grad, "grad"), (m, "exp_avg"), (v, "exp_avg_sq")):
            self._chk(t, nme, (self.n_params,))
        self._packed_tok = None
        self._run("pinn_adam_step", self.lib.pinn_adam_step, _ptr(params), _ptr(grad), _ptr(m), _ptr(v), self.n_params, step, lr, beta1,
                                      beta2, eps)
