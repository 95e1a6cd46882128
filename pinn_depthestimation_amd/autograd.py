"""autograd glue between torch's graph and the HIP engine.

The reference differentiates the network N times in reverse mode
(physics.py:6-15) and then differentiates that again (train.py:191).  Here the
network's input-Jacobian comes from ONE forward-mode pass (pinn_forward_jet) and
parameter gradients from ONE hand-written reverse sweep (pinn_jet_backward, or the
fused pinn_residual_loss_grad); torch only threads the pieces together:

  DNN.forward(X)        -> _PlainForward      Y = net(X)                  (pinn_forward)
                           _AttachInputs      identity on Y whose backward hands
                                              d(.)/dX = sum_c gY[:,c]*dY[j][:,c] to the
                                              input columns, built from torch ops on dY
  first compute_gradient-> _JetTangents       dY = d Y / d X[:, grad cols] (pinn_forward_jet),
                                              created lazily, graph-connected to the params
  loss.backward()       -> _PlainForward.backward / _JetTangents.backward = pinn_jet_backward

so `compute_gradient(pred, var)` keeps its signature and any user expression
(h*U, (h+z)*u ...) goes through ordinary autograd, while physics.<residual> can
recognise outputs of DNN.forward (JetTensor) and call the fused kernel instead.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from ._lib import PinnError
from .engine import ACTIVATION_OF_INIT, Engine, NetDesc, ResidualSpec

_ENGINES = {}


def engine_for(model, grad_cols: Sequence[int], device, drop=(0.0, 0)) -> Engine:
    """The engine for this geometry (cached), with the dropout seed of the CURRENT forward pass set on it:
    drop = (p, seed); every kernel call of one pass, forward and reverse, runs under the pass's own seed."""
    key = (tuple(model.layer_sizes), tuple(grad_cols), model.init_type, str(device), float(drop[0]))
    if key not in _ENGINES:
        desc = NetDesc.from_layers(model.layer_sizes, grad_cols, ACTIVATION_OF_INIT[model.init_type],
                                   dropout_p=float(drop[0]))
        _ENGINES[key] = Engine(desc, device)
    eng = _ENGINES[key]
    eng.dropout_seed = int(drop[1])
    return eng


def _split_flat(model, flat_grad: torch.Tensor):
    out, off = [], 0
    for p in model._ordered_params():
        n = p.numel()
        out.append(flat_grad[off:off + n].view_as(p))
        off += n
    return out


class _PlainForward(torch.autograd.Function):
    """Y = net(X) with X treated as data; backward = reverse sweep for d/d theta."""

    @staticmethod
    def forward(ctx, model, X, drop, *params):
        eng = engine_for(model, (), X.device, drop)
        ctx.model, ctx.drop = model, drop
        ctx.save_for_backward(X)
        return eng.forward(model.flat_params(), X)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gY):
        (X,) = ctx.saved_tensors
        model = ctx.model
        grad = torch.zeros_like(model.flat_params())
        engine_for(model, (), X.device, ctx.drop).jet_backward(model.flat_params(), X, gY.contiguous(), None, grad)
        return (None, None, None, *_split_flat(model, grad))


class _JetTangents(torch.autograd.Function):
    """dY[j] = d net(X) / d X[:, grad_cols[j]]  (k, N, d_out), one forward-mode pass."""

    @staticmethod
    def forward(ctx, model, X, grad_cols, drop, *params):
        eng = engine_for(model, grad_cols, X.device, drop)
        ctx.model, ctx.drop, ctx.grad_cols = model, drop, tuple(grad_cols)
        ctx.save_for_backward(X)
        _, dY = eng.forward_jet(model.flat_params(), X)
        return dY

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gdY):
        (X,) = ctx.saved_tensors
        model = ctx.model
        grad = torch.zeros_like(model.flat_params())
        engine_for(model, ctx.grad_cols, X.device, ctx.drop).jet_backward(model.flat_params(), X, None, gdY.contiguous(), grad)
        return (None, None, None, None, *_split_flat(model, grad))


class JetHandle:
    """What one DNN.forward call knows: the model, the data matrix and which input tensors
    fed which differentiated column."""

    def __init__(self, model, X_data: torch.Tensor, X_graph: torch.Tensor, grad_cols, sources, drop=(0.0, 0)):
        self.model, self.X, self.X_graph = model, X_data, X_graph
        self.drop = drop                  # (p, seed) of this forward pass: its tangents / fused residual reuse the mask
        self.grad_cols: Tuple[int, ...] = tuple(grad_cols)
        self.sources = sources            # per grad col: ("leaf", tensor) | ("node", grad_fn) | None
        self._dY: Optional[torch.Tensor] = None

    def tangents(self) -> torch.Tensor:
        stale = self._dY is not None and torch.is_grad_enabled() and not self._dY.requires_grad and \
            any(p.requires_grad for p in self.model._ordered_params())
        if self._dY is None or stale:   # never reuse tangents that were built outside the graph
            self._dY = _JetTangents.apply(self.model, self.X, self.grad_cols, self.drop, *self.model._ordered_params())
        return self._dY

    def direction_of(self, var: torch.Tensor) -> Optional[int]:
        """Index j such that `var` is the input tensor behind X[:, grad_cols[j]], else None."""
        for j, src in enumerate(self.sources):
            if src is None:
                continue
            kind, obj = src
            if kind == "leaf" and obj is var:
                return j
            if kind == "node" and var.grad_fn is not None and var.grad_fn is obj:
                return j
        return None


class _AttachInputs(torch.autograd.Function):
    """Identity on Y that routes input-gradients through the forward-mode tangents."""

    @staticmethod
    def forward(ctx, Y, X_graph, handle):
        ctx.handle = handle
        return Y.view_as(Y)

    @staticmethod
    def backward(ctx, gY):
        h = ctx.handle
        gX = None
        if ctx.needs_input_grad[1]:
            dY = h.tangents()                                  # (k, N, d_out), carries the param graph
            cols = (gY.unsqueeze(0) * dY).sum(-1)              # (k, N)
            gX = torch.zeros(h.X.shape, dtype=gY.dtype, device=gY.device)
            gX = gX.index_copy(1, torch.tensor(h.grad_cols, device=gY.device), cols.t())
        return gY, gX, None


class JetTensor(torch.Tensor):
    """Output of DNN.forward: an ordinary tensor that remembers its JetHandle and which
    output columns it holds, so physics.<residual>(…) can find the fused path.  Any
    operation other than column slicing returns a plain torch.Tensor."""

    @staticmethod
    def wrap(t: torch.Tensor, handle: JetHandle, cols: Tuple[int, ...]) -> "JetTensor":
        r = t.as_subclass(JetTensor)
        r._pinn_handle, r._pinn_cols = handle, cols
        return r

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if func is torch.Tensor.__getitem__ and isinstance(args[0], JetTensor) and isinstance(out, torch.Tensor):
            cols = _sliced_cols(args[0], args[1], out)
            if cols is not None:
                return JetTensor.wrap(out, args[0]._pinn_handle, cols)
        return out


def _sliced_cols(src: JetTensor, idx, out) -> Optional[Tuple[int, ...]]:
    """predictions[:, i:i+1] (train.py:138,150) -> the output columns kept; None if rows were touched."""
    handle = getattr(src, "_pinn_handle", None)
    if handle is None or src.dim() != 2 or out.dim() != 2 or out.shape[0] != src.shape[0]:
        return None
    if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None)):
        return None
    col = idx[1]
    base = src._pinn_cols
    if isinstance(col, slice):
        picked = tuple(base[i] for i in range(*col.indices(len(base))))
        return picked if len(picked) == out.shape[1] else None
    return None


def _edge_shape(node, idx):
    """Shape of the tensor an autograd edge (node, output index) carries, or None if torch does not say."""
    v = getattr(node, "variable", None)
    if isinstance(v, torch.Tensor):
        return tuple(v.shape)
    meta = getattr(node, "_input_metadata", None)
    try:
        return tuple(meta[idx].shape)
    except Exception:
        return None


def _is_cat_of_columns(fn, x: torch.Tensor) -> bool:
    """Is `fn` the backward node of torch.cat([... (N,1) columns ...], dim=-1) that produced x?
    Decided STRUCTURALLY, not by the node's class name (a torch internal that may be renamed): the node
    has exactly one input edge per column of x, and every edge that carries a gradient comes from an
    (N, 1) tensor.  No other single operation assembles an (N, d_in) matrix from (N, 1) operands — an
    elementwise op on an (N, d_in) leaf has (N, d_in) operands.  A recorded concatenation axis, where
    torch exposes one, must be the last."""
    d_in = x.shape[1]
    if fn is None or len(fn.next_functions) != d_in:
        return False
    dim = getattr(fn, "_saved_dim", None)
    if dim is not None and dim not in (-1, x.dim() - 1, 2 ** 64 - 1):   # (-1 is stored as an unsigned wrap)
        return False
    live = [(n, i) for n, i in fn.next_functions if n is not None]
    if not live:
        return False
    return all(_edge_shape(n, i) == (x.shape[0], 1) for n, i in live)


def _leaf_of(node):
    """The leaf tensor behind an accumulate-grad node (identified by its `.variable`, not its name)."""
    v = getattr(node, "variable", None)
    return v if isinstance(v, torch.Tensor) else None


def _sniff_sources(model, x: torch.Tensor):
    """Which X columns are differentiated, and which user tensor sits behind each one.
    train.py:86-88,144-148: the inputs are (N,1) tensors joined by torch.cat(dim=-1)."""
    d_in = x.shape[1]
    override = getattr(model, "_grad_cols_override", None)
    fn = x.grad_fn
    if _is_cat_of_columns(fn, x):
        cols, sources = [], []
        for i, (node, _) in enumerate(fn.next_functions):
            if node is None:
                continue
            cols.append(i)
            leaf = _leaf_of(node)
            sources.append(("leaf", leaf) if leaf is not None else ("node", node))
        if override is not None and tuple(cols) != tuple(override):
            raise PinnError(f"set_grad_columns({override}) disagrees with the inputs' requires_grad ({cols})")
        return tuple(cols), sources
    if override is not None:
        return tuple(override), [None] * len(override)
    if x.is_leaf and d_in <= 3:
        # test.py:62-76 style: ONE leaf matrix with requires_grad — every column is a differentiated input
        # (3 = the engine's PINN_MAX_DIRS); wider leaf matrices must name their columns
        return tuple(range(d_in)), [None] * d_in
    raise PinnError(
        "cannot tell which input columns are differentiated: build the input with "
        "torch.cat([... (N,1) columns ...], dim=-1) as train.py:148 does, or call "
        "model.set_grad_columns([...])")


def dnn_forward(model, x: torch.Tensor, drop=(0.0, 0)):
    if x.dim() != 2 or x.shape[1] != model.layer_sizes[0]:
        raise PinnError(f"input has shape {tuple(x.shape)}, expected (N, {model.layer_sizes[0]})")
    Xd = x.detach().to(torch.float32).contiguous()
    params = model._ordered_params()
    Y = _PlainForward.apply(model, Xd, drop, *params)
    if not (x.requires_grad and torch.is_grad_enabled()):
        return Y
    grad_cols, sources = _sniff_sources(model, x)
    if len(grad_cols) == 0:
        return Y
    handle = JetHandle(model, Xd, x, grad_cols, sources, drop)
    Y2 = _AttachInputs.apply(Y, x, handle)
    return JetTensor.wrap(Y2, handle, tuple(range(Y2.shape[1])))


# ---- fused residual -------------------------------------------------------------------------


class _FusedResidual(torch.autograd.Function):
    """loss = sum_t mean(field_t^2) and d loss / d theta from ONE kernel call
    (pinn_residual_loss_grad); backward only scales the stored gradient."""

    @staticmethod
    def forward(ctx, model, handle, spec, *params):
        eng = engine_for(model, handle.grad_cols, handle.X.device, handle.drop)
        flat = model.flat_params()
        N = handle.X.shape[0]
        grad = torch.zeros_like(flat)
        if spec.name == "continuity_only":
            xcol = handle.grad_cols[spec.dir_of[0]]
            cnt = (handle.X[:, xcol] < spec.threshold).sum().to(torch.float32)
            scale = torch.stack([torch.tensor(1.0 / N, device=flat.device), 1.0 / cnt,
                                 torch.tensor(0.0, device=flat.device)])      # 1/0 -> inf -> NaN, as the
        else:                                                                 # reference's mean of empty
            scale = torch.full((spec.n_terms,), 1.0 / N, device=flat.device)
        sums = eng.residual_loss_grad(spec, scale, flat, handle.X, grad)
        ctx.model, ctx.grad = model, grad
        return (sums * scale).sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return (None, None, None, *_split_flat(ctx.model, ctx.grad * g))


def fused_residual(name: str, in_vars: Sequence[torch.Tensor], out_vars: Sequence[torch.Tensor]):
    """Return the fused loss tensor if every argument is recognisably (input column of /
    output column of) one DNN.forward call, else None."""
    handle = None
    out_col: List[int] = []
    for o in out_vars:
        h = getattr(o, "_pinn_handle", None)
        cols = getattr(o, "_pinn_cols", None)
        if not isinstance(o, JetTensor) or h is None or cols is None or len(cols) != 1:
            return None
        if handle is None:
            handle = h
        elif h is not handle:
            return None
        out_col.append(cols[0])
    dir_of: List[int] = []
    for v in in_vars:
        j = handle.direction_of(v)
        if j is None:
            return None
        dir_of.append(j)
    spec = ResidualSpec(name, tuple(out_col), tuple(dir_of))
    model = handle.model
    return _FusedResidual.apply(model, handle, spec, *model._ordered_params())
