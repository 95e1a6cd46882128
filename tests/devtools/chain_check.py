"""Development check of the bf16-mode chain kernels against the fp64 oracle on several shapes (GPU box)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import pinn_oracle as O
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd._lib import ENGINE_WIDE, PREC_BF16, PREC_F32
from tests.golden_util import oracle_loss_and_grad, rel_l2

CASES = {
    "ns_2x256": (3, 4, 2, 256, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "ns_12x256": (3, 4, 12, 256, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "ns_3x128": (3, 4, 3, 128, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "ns_1x256": (3, 4, 1, 256, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v")),
    "pe_3x100": (2, 6, 3, 100, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    "co_4x200": (2, 3, 4, 200, (0, 1), "continuity_only", ("x", "y"), ("U", "V", "h")),
}
which = sys.argv[1:] or list(CASES)
for name in which:
    d_in, d_out, L, W, gc, res, inn, outn = CASES[name]
    for N in (1237, 70):
        g = torch.Generator().manual_seed(4321)
        params = O.init_params(O.layer_sizes(d_in, L, W, d_out), "xavier", g)
        if res == "physics_equation":
            params[-1][outn.index("h")] = 0.75; params[-1][outn.index("eta_mean")] = 0.0
        X = torch.rand(N, d_in, generator=g) * 2 - 1
        if res == "continuity_only": X[:, 0] = X[:, 0] * 40
        desc = NetDesc(d_in, d_out, L, W, gc, engine=ENGINE_WIDE)
        spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
        l64, g64 = oracle_loss_and_grad(params, X, res, inn, outn, desc.grad_cols, torch.float64)
        flat, Xd = O.flatten(params).cuda(), X.cuda().contiguous()
        if res == "continuity_only":
            cnt = float((X[:, 0] < 25.5).sum()); scale = torch.tensor([1.0 / N, 1.0 / cnt, 0.0]).cuda()
        else:
            scale = torch.full((spec.n_terms,), 1.0 / N).cuda()
        for prec in (PREC_F32, PREC_BF16):
            eng = Engine(desc.with_(precision=prec))
            grad = torch.zeros(desc.n_params, device="cuda")
            sums = eng.residual_loss_grad(spec, scale, flat, Xd, grad)
            torch.cuda.synchronize()
            loss = float((sums.double() * scale.double()).sum())
            el, eg = abs(loss - float(l64)) / abs(float(l64)), rel_l2(grad.cpu(), g64)
            # per-layer gradient error
            offs, per = 0, []
            for i in range(L + 1):
                nw = params[2 * i].numel(); nb = params[2 * i + 1].numel()
                per.append("%.1e/%.1e" % (rel_l2(grad[offs:offs + nw].cpu(), g64[offs:offs + nw]), rel_l2(grad[offs + nw:offs + nw + nb].cpu(), g64[offs + nw:offs + nw + nb])))
                offs += nw + nb
            Y = eng.forward(flat, Xd)
            Yo = O.mlp_forward([p.double() for p in params], X.double())
            ey = float((Y.cpu().double() - Yo).abs().max())
            print(f"{name:10s} N={N:5d} {'bf16' if prec else 'f32 '}: loss err {el:.2e} grad err {eg:.2e} fwd maxabs {ey:.2e} | per layer W/b: {' '.join(per)}", flush=True)
