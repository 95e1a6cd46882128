"""The C-ABI's host side without a GPU: every entry point must validate its arguments and return an error code with a
message BEFORE anything is launched.  These tests pass NULL / inconsistent arguments straight through ctypes; they are
also what tools/asan_abi_check.sh runs against the AddressSanitizer + UBSan host build of the library."""
import ctypes as C

import pytest

from pinn_depthestimation_amd import NetDesc, ResidualSpec, _lib

NULL = None
OK, INVALID, UNSUPPORTED, WORKSPACE = 0, -1, -2, -3


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def err(lib):
    return lib.pinn_last_error().decode()


def test_abi_descriptor_validation(lib):
    need, cnt = C.c_int64(), C.c_int64()
    assert lib.pinn_param_count(NULL, C.byref(cnt)) == INVALID and "desc is NULL" in err(lib)
    good = NetDesc(3, 4, 8, 64, (0, 1, 2))
    assert lib.pinn_param_count(C.byref(good.c_struct()), NULL) == INVALID
    for bad, what in ((NetDesc(0, 4, 8, 64), "bad network shape"), (NetDesc(3, 4, 0, 64), "bad network shape"),
                      (NetDesc(3, 4, 8, 64, (0, 5)), "outside the 3 input columns"),
                      (NetDesc(3, 4, 8, 64, (), activation=7), "invalid activation"),
                      (NetDesc(3, 4, 8, 64, (), precision=9), "invalid precision"),
                      (NetDesc(3, 4, 8, 64, (), engine=17), "invalid engine")):
        assert lib.pinn_query_workspace(C.byref(bad.c_struct()), 10, C.byref(need)) == INVALID, what
        assert what in err(lib), (what, err(lib))
    d = good.c_struct(); d.k = 4
    assert lib.pinn_query_workspace(C.byref(d), 10, C.byref(need)) == INVALID and "k=4" in err(lib)
    assert lib.pinn_query_workspace(C.byref(good.c_struct()), -1, C.byref(need)) == INVALID
    assert lib.pinn_query_workspace(C.byref(good.c_struct()), 10, NULL) == INVALID


def test_abi_engine_selection_and_workspace_sizes(lib):
    need = C.c_int64()
    sizes = {}
    for name, desc in (("fused", NetDesc(3, 4, 8, 64, (0, 1, 2))), ("fused_tile", NetDesc(3, 4, 8, 64, (0, 1, 2), engine=4)),
                       ("fused_coop", NetDesc(3, 4, 8, 64, (0, 1, 2), engine=5)), ("generic", NetDesc(3, 4, 8, 64, (0, 1, 2), engine=1)),
                       ("wide_f32", NetDesc(3, 4, 12, 256, (0, 1, 2))), ("wide_bf16", NetDesc(3, 4, 12, 256, (0, 1, 2), precision=1)),
                       ("narrow100x20", NetDesc(2, 3, 100, 20, (0, 1))), ("wide_k1_generic", NetDesc(3, 4, 2, 300, (0, 1, 2)))):
        assert lib.pinn_query_workspace(C.byref(desc.c_struct()), 4096, C.byref(need)) == OK, (name, err(lib))
        sizes[name] = need.value
        assert need.value > 0
    assert sizes["wide_bf16"] != sizes["wide_f32"]
    # shapes an engine cannot serve are refused with the engine named, not silently rerouted
    for desc, what in ((NetDesc(3, 4, 8, 128, (0, 1, 2), engine=2), "fused engine does not support"),
                       (NetDesc(3, 4, 8, 64, (0, 1, 2), engine=3), "wide engine does not support"),
                       (NetDesc(3, 4, 8, 32, (0, 1, 2), engine=5), "fused engine does not support"),      # coop: width 33..64 only
                       (NetDesc(3, 4, 8, 64, (0, 1, 2), precision=1), "bf16 is implemented on the wide engine"),
                       (NetDesc(3, 4, 60, 256, (0, 1, 2), precision=1), "bf16 is implemented on the wide engine")):
        assert lib.pinn_query_workspace(C.byref(desc.c_struct()), 100, C.byref(need)) == UNSUPPORTED, what
        assert what in err(lib), (what, err(lib))
    # k = 1 networks: forward on the fused engine, gradient on the generic one -> the workspace serves both
    k1 = NetDesc(3, 4, 4, 32, (1,))
    assert lib.pinn_query_workspace(C.byref(k1.c_struct()), 4096, C.byref(need)) == OK
    gen = C.c_int64()
    assert lib.pinn_query_workspace(C.byref(k1.with_(engine=1).c_struct()), 4096, C.byref(gen)) == OK
    assert need.value >= gen.value


def test_abi_call_argument_validation_returns_before_any_launch(lib):
    desc = NetDesc(3, 4, 8, 64, (0, 1, 2)).c_struct()
    spec = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), (0, 1, 2), ("h", "z", "u", "v")).c_struct()
    fake = C.c_void_p(0x1000)        # never dereferenced: validation fails first
    assert lib.pinn_forward(C.byref(desc), NULL, fake, 10, fake, fake, 1 << 20, NULL) == INVALID
    assert lib.pinn_forward(C.byref(desc), fake, fake, -5, fake, fake, 1 << 20, NULL) == INVALID
    assert lib.pinn_forward(C.byref(desc), fake, NULL, 0, NULL, NULL, 0, NULL) == OK               # N = 0: nothing to do
    assert lib.pinn_forward_jet(C.byref(desc), fake, fake, 10, fake, NULL, fake, 1 << 20, NULL) == INVALID
    d0 = NetDesc(3, 4, 8, 64, ()).c_struct()
    assert lib.pinn_forward_jet(C.byref(d0), fake, fake, 10, fake, fake, fake, 1 << 20, NULL) == INVALID and "k >= 1" in err(lib)
    assert lib.pinn_residual_loss_grad(C.byref(desc), NULL, fake, fake, fake, 10, fake, fake, fake, 1 << 20, NULL) == INVALID
    assert "spec is NULL" in err(lib)
    bad = ResidualSpec("Navier_Stokes", (0, 1, 2, 9), (0, 1, 2)).c_struct()
    assert lib.pinn_residual_loss_grad(C.byref(desc), C.byref(bad), fake, fake, fake, 10, fake, fake, fake, 1 << 20, NULL) == INVALID
    assert "out_col[3]=9" in err(lib)
    bad = ResidualSpec("Navier_Stokes", (0, 1, 2, 3), (0, 1, 3)).c_struct()
    assert lib.pinn_residual_loss(C.byref(desc), C.byref(bad), fake, fake, 10, fake, fake, 1 << 20, NULL) == INVALID
    assert "tangent directions" in err(lib)
    spec.residual_id = 77
    assert lib.pinn_residual_loss(C.byref(desc), C.byref(spec), fake, fake, 10, fake, fake, 1 << 20, NULL) == INVALID
    assert "unknown residual_id" in err(lib)
    spec.residual_id = 1
    assert lib.pinn_residual_loss_grad(C.byref(desc), C.byref(spec), NULL, fake, fake, 10, fake, fake, fake, 1 << 20, NULL) == INVALID
    oc = (C.c_int32 * 2)(0, 7)
    assert lib.pinn_mse_loss_grad(C.byref(desc), fake, fake, fake, 10, 2, oc, fake, fake, fake, fake, 1 << 20, NULL) == INVALID
    assert "out_col[1]=7" in err(lib)
    assert lib.pinn_mse_loss_grad(C.byref(desc), fake, fake, fake, 10, 0, oc, fake, fake, fake, fake, 1 << 20, NULL) == INVALID
    assert lib.pinn_mse_loss_grad(C.byref(desc), fake, fake, fake, 10, 9, oc, fake, fake, fake, fake, 1 << 20, NULL) == INVALID
    oc = (C.c_int32 * 2)(0, 1)
    assert lib.pinn_residual_mse_split_loss_grad(C.byref(desc), C.byref(spec), fake, fake, 2, oc, fake, fake, fake, 10, 11,
                                                 fake, fake, fake, fake, 1 << 20, NULL) == INVALID and "exceeds N" in err(lib)
    assert lib.pinn_residual_mse_split_loss_grad(C.byref(desc), C.byref(spec), fake, fake, 2, oc, fake, fake, fake, 10, -1,
                                                 fake, fake, fake, fake, 1 << 20, NULL) == INVALID
    assert lib.pinn_adam_step(NULL, fake, fake, fake, 10, 1, 1e-4, 0.9, 0.999, 1e-8, NULL) == INVALID
    assert lib.pinn_adam_step(fake, fake, fake, fake, 10, 0, 1e-4, 0.9, 0.999, 1e-8, NULL) == INVALID     # step is 1-based
    assert lib.pinn_adam_step(fake, fake, fake, fake, 0, 1, 1e-4, 0.9, 0.999, 1e-8, NULL) == OK
    assert lib.pinn_nanminmax_f64(NULL, 10, fake, fake, 1 << 20, NULL) == INVALID
    assert lib.pinn_stage_workspace_bytes(0, 5, 1, 1) == -1 and lib.pinn_stage_workspace_bytes(81, 261, 1, 1) > 0
    g = (C.c_void_p * 2)(0x1000, 0)
    assert lib.pinn_stage_grid_columns(g, 2, 9, 7, 1, 1, fake, fake, fake, fake, 1 << 20, NULL) == INVALID and "grid 1 is NULL" in err(lib)
    assert lib.pinn_stage_grid_columns(g, 17, 9, 7, 1, 1, fake, fake, fake, fake, 1 << 20, NULL) == INVALID
    # a workspace that is too small is reported with the size that is needed (the engines check before launching)
    need = C.c_int64()
    lib.pinn_query_workspace(C.byref(desc), 1000, C.byref(need))
    assert lib.pinn_forward(C.byref(desc), fake, fake, 1000, fake, fake, need.value - 1, NULL) == WORKSPACE
    assert str(need.value) in err(lib)


def test_abi_folded_adam_iteration_validation(lib):
    """pinn_loss_grad_adam_step / pinn_adam_loop: argument checks and the refusal of requests that are not one pass of the
    fused engine happen on the host, before anything is launched (fake non-NULL device pointers are never dereferenced)."""
    fake = C.c_void_p(4096)
    good = NetDesc(3, 4, 8, 64, (0, 1, 2))
    spec = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), good.grad_cols, ("h", "z", "u", "v"))
    st = _lib.PinnAdamState(fake, fake, 1, 1e-3, 0.9, 0.999, 1e-8, 0, 0, None, None)
    oc = (C.c_int32 * 1)(0)

    def call(desc, adam, n_cols=0, n_res=100, N=100):
        return lib.pinn_loss_grad_adam_step(C.byref(desc.c_struct()), C.byref(spec.c_struct()), fake, fake, n_cols, oc, fake,
                                            fake, fake, N, n_res, fake, fake, fake, adam, fake, 1 << 30, None)
    assert call(good, None) == INVALID
    bad = _lib.PinnAdamState(None, fake, 1, 1e-3, 0.9, 0.999, 1e-8, 0, 0, None, None)
    assert call(good, C.byref(bad)) == INVALID
    bad = _lib.PinnAdamState(fake, fake, 0, 1e-3, 0.9, 0.999, 1e-8, 0, 0, None, None)       # steps count from 1
    assert call(good, C.byref(bad)) == INVALID
    assert call(good, C.byref(st), n_cols=0, n_res=50) == INVALID and "n_res must equal N" in err(lib)
    bad = _lib.PinnAdamState(fake, fake, 1, 1e-3, 0.9, 0.999, 1e-8, 0, 3, None, None)         # loss rows without buffers
    assert call(good, C.byref(bad)) == INVALID and "loss_rows" in err(lib)
    # the wide engine (and the generic one) are refused, nothing launched
    for desc in (NetDesc(3, 4, 12, 256, (0, 1, 2)), NetDesc(3, 4, 8, 64, (0, 1, 2), engine=1)):
        assert call(desc, C.byref(st)) == UNSUPPORTED and "one-pass request on the fused engine" in err(lib)
    lr = (C.c_double * 2)(1e-3, 1e-3)
    args = (C.byref(good.c_struct()), C.byref(spec.c_struct()), fake, fake, 0, oc, fake, fake, fake, 100, 100, fake, fake, fake)
    assert lib.pinn_adam_loop(*args, C.byref(st), 2, None, fake, 1 << 30, None) == INVALID
    assert lib.pinn_adam_loop(*args, None, 2, lr, fake, 1 << 30, None) == INVALID
    assert lib.pinn_adam_loop(*args, C.byref(st), 0, lr, fake, 1 << 30, None) == OK            # zero iterations: nothing to do


def test_abi_workspace_serves_the_calls_that_drop_the_tangents(lib):
    """pinn_forward / pinn_mse_loss_grad run a k > 0 network as a k = 0 one, which AUTO may give to ANOTHER engine:
    k = 1 at width 65..256 has its jets on the generic kernels and its plain forward on the wide engine (round-3
    defect found by tests/test_sweep_gpu.py: the query sized the generic kernels' workspace only)."""
    need, plain = C.c_int64(), C.c_int64()
    for shape in ((2, 4, 2, 72), (1, 6, 12, 72), (1, 5, 5, 100), (3, 4, 4, 32), (3, 4, 8, 64), (3, 4, 12, 256)):
        d_in, d_out, L, W = shape
        for k in (1, 2, 3):
            if k > d_in: continue
            for N in (15, 700, 9600):
                jet = NetDesc(d_in, d_out, L, W, tuple(range(k)))
                assert lib.pinn_query_workspace(C.byref(jet.c_struct()), N, C.byref(need)) == OK, err(lib)
                assert lib.pinn_query_workspace(C.byref(NetDesc(d_in, d_out, L, W, ()).c_struct()), N, C.byref(plain)) == OK
                assert need.value >= plain.value, (shape, k, N, need.value, plain.value)
