"""ctypes binding of libpinn_hip.so (include/pinn_hip.h).  No torch extension headers,
no pybind: the shared library is a plain C-ABI and this file is the whole binding."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PINN_HIP_LIB", os.path.join(_HERE, "libpinn_hip.so"))  # override: kernel experiments
CSRC = os.path.join(_HERE, "csrc")

PINN_MAX_DIRS = 3
PINN_MAX_ROLES = 8

ACT_TANH, ACT_LEAKY_RELU = 0, 1
ENGINE_AUTO, ENGINE_GENERIC, ENGINE_FUSED, ENGINE_WIDE = 0, 1, 2, 3
ENGINE_FUSED_TILE, ENGINE_FUSED_COOP, ENGINE_FUSED_BATCH = 4, 5, 6     # sub-values of ENGINE_FUSED: force one of its kernels (pinn_hip.h)
ABI_VERSION = 3
PREC_F32, PREC_BF16 = 0, 1

RES_NAVIER_STOKES, RES_PHYSICS_EQUATION, RES_CONTINUITY_FTEMP, RES_CONTINUITY_ONLY = 1, 2, 3, 4
RES_TERMS = {RES_NAVIER_STOKES: 3, RES_PHYSICS_EQUATION: 3, RES_CONTINUITY_FTEMP: 1, RES_CONTINUITY_ONLY: 3}


class PinnDesc(C.Structure):
    _fields_ = [
        ("d_in", C.c_int32), ("d_out", C.c_int32), ("n_hidden", C.c_int32), ("width", C.c_int32),
        ("k", C.c_int32), ("dir_col", C.c_int32 * PINN_MAX_DIRS),
        ("activation", C.c_int32), ("engine", C.c_int32), ("precision", C.c_int32),
        ("dropout_p", C.c_float), ("dropout_seed", C.c_uint32),
    ]


class PinnResidualSpec(C.Structure):
    _fields_ = [
        ("residual_id", C.c_int32), ("out_col", C.c_int32 * PINN_MAX_ROLES),
        ("dir_of", C.c_int32 * PINN_MAX_DIRS), ("flags", C.c_int32), ("param", C.c_float * 4),
    ]


class PinnAdamState(C.Structure):
    _fields_ = [
        ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_int64), ("lr", C.c_double), ("beta1", C.c_double),
        ("beta2", C.c_double), ("eps", C.c_double), ("packed_valid", C.c_int32), ("n_loss_rows", C.c_int32),
        ("loss_rows", C.c_void_p), ("losses", C.c_void_p),
    ]


ERR_UNSUPPORTED = -2


class PinnError(RuntimeError):
    pass


_P = C.c_void_p
_SIGNATURES = {
    "pinn_version": (C.c_int32, []),
    "pinn_last_error": (C.c_char_p, []),
    "pinn_dropout_keep": (C.c_int32, [C.c_uint32, C.c_int32, C.c_int32, C.c_int64, C.c_float]),
    "pinn_param_count": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(C.c_int64)]),
    "pinn_query_workspace": (C.c_int32, [C.POINTER(PinnDesc), C.c_int64, C.POINTER(C.c_int64)]),
    "pinn_forward": (C.c_int32, [C.POINTER(PinnDesc), _P, _P, C.c_int64, _P, _P, C.c_int64, _P]),
    "pinn_forward_jet": (C.c_int32, [C.POINTER(PinnDesc), _P, _P, C.c_int64, _P, _P, _P, C.c_int64, _P]),
    "pinn_jet_backward": (C.c_int32, [C.POINTER(PinnDesc), _P, _P, C.c_int64, _P, _P, _P, _P, C.c_int64, _P]),
    "pinn_residual_loss": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(PinnResidualSpec), _P, _P, C.c_int64, _P,
                                       _P, C.c_int64, _P]),
    "pinn_residual_loss_grad": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(PinnResidualSpec), _P, _P, _P,
                                            C.c_int64, _P, _P, _P, C.c_int64, _P]),
    "pinn_mse_loss_grad": (C.c_int32, [C.POINTER(PinnDesc), _P, _P, _P, C.c_int64, C.c_int32,
                                       C.POINTER(C.c_int32), _P, _P, _P, _P, C.c_int64, _P]),
    "pinn_residual_mse_loss_grad": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(PinnResidualSpec), _P, _P, C.c_int32,
                                                C.POINTER(C.c_int32), _P, _P, _P, C.c_int64, _P, _P, _P, _P,
                                                C.c_int64, _P]),
    "pinn_residual_mse_split_loss_grad": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(PinnResidualSpec), _P, _P, C.c_int32,
                                                      C.POINTER(C.c_int32), _P, _P, _P, C.c_int64, C.c_int64, _P, _P, _P,
                                                      _P, C.c_int64, _P]),
    "pinn_lbfgs_push": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P]),
    "pinn_lbfgs_direction": (C.c_int32, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, C.c_double, _P,
                                         _P, _P, _P, _P]),
    "pinn_nanminmax_f64": (C.c_int32, [_P, C.c_int64, _P, _P, C.c_int64, _P]),
    "pinn_stage_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "pinn_stage_grid_columns": (C.c_int32, [C.POINTER(_P), C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P, _P,
                                            C.c_int64, _P]),
    "pinn_adam_step": (C.c_int32, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                   C.c_double, _P]),
    "pinn_loss_grad_adam_step": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(PinnResidualSpec), _P, _P, C.c_int32,
                                             C.POINTER(C.c_int32), _P, _P, _P, C.c_int64, C.c_int64, _P, _P, _P,
                                             C.POINTER(PinnAdamState), _P, C.c_int64, _P]),
    "pinn_adam_loop": (C.c_int32, [C.POINTER(PinnDesc), C.POINTER(PinnResidualSpec), _P, _P, C.c_int32,
                                   C.POINTER(C.c_int32), _P, _P, _P, C.c_int64, C.c_int64, _P, _P, _P,
                                   C.POINTER(PinnAdamState), C.c_int32, C.POINTER(C.c_double), _P, C.c_int64, _P]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libpinn_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode != 0:
        raise PinnError("building libpinn_hip.so failed (see output above)")
    return LIB_PATH


def load():
    """Load the HIP library.  There is no CPU fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PinnError(
            f"{LIB_PATH} is missing: the MI355X engine has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            f"{CSRC}`) first; this package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.pinn_version() != ABI_VERSION:
        raise PinnError(f"{LIB_PATH} has ABI version {lib.pinn_version()}, this package binds version {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pinn_last_error()
        raise PinnError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
