"""Seeded random sweep over the drop-in Python face (dnn.DNN + physics.* through torch autograd, the way train.py:86-157
is written): random input / output column orders, extra columns, which inputs carry requires_grad, an optional
fidelity term on the same or on other columns, widths on every engine — against the CPU oracle in float64 (checker
only).  The hand-written drop-in tests all use the reference configs' own column order; the C-ABI sweep
(test_sweep_gpu.py) found an order-dependent defect there, this is the same net cast over the autograd boundary
(graph sniffing of the differentiated columns, role mapping, gradient hand-back to the Parameters)."""
import os
import random
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "compat"))

from oracle import pinn_oracle as O                                   # noqa: E402  (checker)
from pinn_depthestimation_amd.engine import RESIDUAL_ROLES            # noqa: E402
from tests.golden_util import rel_l2                                  # noqa: E402

pytestmark = pytest.mark.gpu


def draw(seed):
    r = random.Random(seed)
    res = r.choice(sorted(RESIDUAL_ROLES))
    _, out_roles, dir_roles = RESIDUAL_ROLES[res]
    inn = list(dir_roles) + [f"in{i}" for i in range(r.choice([0, 0, 1, 2]))]
    outn = list(out_roles) + [f"out{i}" for i in range(r.choice([0, 0, 1, 2]))]
    r.shuffle(inn); r.shuffle(outn)
    gc = sorted(inn.index(d) for d in dir_roles)
    extra = [i for i in range(len(inn)) if i not in gc]
    if extra and len(gc) < 3 and r.random() < 0.3:
        gc = sorted(gc + [r.choice(extra)])
    L, W = r.choice([1, 2, 4, 8, 10]), r.choice([7, 10, 16, 20, 32, 48, 64, 100, 128])
    N = r.choice([12, 243, 777, 1500])
    fid = r.choice(["none", "same_points", "other_points"])
    return res, inn, outn, gc, L, W, N, fid


@pytest.mark.parametrize("seed", range(7000, 7032))
def test_random_train_py_style_step_against_the_oracle(seed):
    import dnn
    import physics
    res, inn, outn, gc, L, W, N, fid = draw(seed)
    _, out_roles, dir_roles = RESIDUAL_ROLES[res]
    d_in, d_out = len(inn), len(outn)
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = dnn.DNN([d_in] + [W] * L + [d_out], 0.0, "xavier").to("cuda")
    with torch.no_grad():
        model.layers[-1].bias.copy_(torch.rand(d_out, generator=g) * 0.2)
        if res == "physics_equation":
            model.layers[-1].bias[outn.index("h")] = 0.75
            model.layers[-1].bias[outn.index("k")] = 0.5
    X = torch.rand(N, d_in, generator=g, dtype=torch.float64) * 2 - 1
    if res == "continuity_only":
        X[:, inn.index("x")] *= 40
    fcols = sorted(random.Random(seed).sample(range(d_out), random.Random(seed).randint(1, min(3, d_out))))
    fw = [0.5 + 0.25 * j for j in range(len(fcols))]
    Nf = N if fid == "same_points" else 17
    Xf = X if fid == "same_points" else torch.rand(Nf, d_in, generator=g, dtype=torch.float64) * 2 - 1
    T = torch.rand(Nf, len(fcols), generator=g, dtype=torch.float64)

    # ---- the engine, driven like train.py:86-157 ----
    cols = [X[:, i:i + 1].clone().requires_grad_(i in gc).float().to("cuda") for i in range(d_in)]
    pred = model(torch.cat(cols, dim=-1))
    named = {n: pred[:, i:i + 1] for i, n in enumerate(outn)}
    loss = getattr(physics, res)(*[cols[inn.index(d)] for d in dir_roles], *[named[o] for o in out_roles])
    if fid != "none":
        pf = pred if fid == "same_points" else model(Xf.float().to("cuda"))
        Tc = T.float().to("cuda")
        for j, (o, w) in enumerate(zip(fcols, fw)):
            loss = loss + w * F.mse_loss(pf[:, o:o + 1], Tc[:, j:j + 1])
    model.zero_grad()
    loss.backward()
    got = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double().cpu()

    # ---- the oracle, float64 and float32 (its own fp32 noise sets the bar for ill-conditioned residuals) ----
    ref = {}
    for dt in (torch.float64, torch.float32):
        ps = [p.detach().cpu().to(dt).clone().requires_grad_(True) for p in model.parameters()]
        lo = O.residual_loss(ps, X.to(dt), res, [inn.index(d) for d in dir_roles], [outn.index(o) for o in out_roles], gc)
        if fid != "none":
            lo = lo + O.fidelity_loss(ps, Xf.to(dt), T.to(dt), fcols, fw)
        ref[dt] = (float(lo.detach()), O.flat_grad(lo, ps).double())
    (l64, g64), (l32, g32) = ref[torch.float64], ref[torch.float32]
    el, eg = abs(loss.item() - l64) / abs(l64), rel_l2(got, g64)
    nl, ng = abs(l32 - l64) / abs(l64), rel_l2(g32, g64)
    print(f"seed {seed}: {res} in {inn} (grad {gc}) out {outn} {L}x{W} N={N} fid={fid}{fcols if fid != 'none' else ''}: "
          f"loss {el:.1e} (fp32 oracle {nl:.1e}) gradient {eg:.1e} ({ng:.1e})")
    assert el < max(5e-6, 4 * nl)
    assert eg < max(2e-5, 4 * ng)
