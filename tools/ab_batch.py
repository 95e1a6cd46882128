"""ab_batch.py: tile kernel vs batch kernel on the narrow shapes (loss + gradient call only, HIP events), and their
agreement with each other.  python tools/ab_batch.py [points]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd.dnn import init_flat_params
from pinn_depthestimation_amd._lib import ENGINE_FUSED_BATCH, ENGINE_FUSED_TILE

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
SHAPES = {
    "pe10x10": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k")),
    "co100x20": (2, 3, 100, 20, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h")),
    "ns20x20": (4, 4, 20, 20, (0, 1, 2), "Navier_Stokes", ("t", "x", "y", "z0"), ("h", "z", "u", "v")),
}
for name, (d_in, d_out, L, W, gc, res, inn, outn) in SHAPES.items():
    g = torch.Generator().manual_seed(1)
    out = {}
    only = os.environ.get("AB_ONLY")
    for tag, e in (("tile", ENGINE_FUSED_TILE), ("batch", ENGINE_FUSED_BATCH)):
        if only and tag != only:
            continue
        desc = NetDesc(d_in, d_out, L, W, gc, engine=e)
        spec = ResidualSpec.from_names(res, inn, desc.grad_cols, outn)
        params = init_flat_params(desc.layers, "xavier", torch.Generator().manual_seed(3)).cuda()
        if res == "physics_equation":
            params[desc.n_params - d_out + 0] = 0.75; params[desc.n_params - d_out + 3] = 0.0
        X = (torch.rand(N, d_in, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
        eng = Engine(desc)
        grad = torch.zeros(desc.n_params, device="cuda")
        scale = torch.full((spec.n_terms,), 1.0 / N, device="cuda")
        for _ in range(2):
            grad.zero_(); sums = eng.residual_loss_grad(spec, scale, params, X, grad)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            grad.zero_()
            a.record()
            sums = eng.residual_loss_grad(spec, scale, params, X, grad)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        out[tag] = (ts[len(ts) // 2], sums.clone(), grad.clone(), ts[0], ts[-1])
    if only:
        print(f"{name}: N={N} {only} {out[only][0]:.3f} ms [{out[only][3]:.3f}..{out[only][4]:.3f}]", flush=True)
        continue
    rel = float((out["tile"][2] - out["batch"][2]).norm() / out["tile"][2].norm())
    rl = float(((out["tile"][1] - out["batch"][1]).abs() / out["tile"][1].abs().clamp_min(1e-30)).max())
    print(f"{name}: N={N} tile {out['tile'][0]:.3f} ms [{out['tile'][3]:.3f}..{out['tile'][4]:.3f}]  batch {out['batch'][0]:.3f} ms [{out['batch'][3]:.3f}..{out['batch'][4]:.3f}]  grad rel diff {rel:.2e}  sums rel diff {rl:.2e}", flush=True)
