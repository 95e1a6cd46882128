"""Development check: forward jets (values + tangents) of the bf16-mode chain forward against the fp32 engine,
per quantity and per point position inside a 16-point tile (GPU box)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import pinn_oracle as O
from pinn_depthestimation_amd import Engine, NetDesc
from pinn_depthestimation_amd._lib import ENGINE_WIDE, PREC_BF16, PREC_F32

CASES = {"ns_2x256": (3, 4, 2, 256, (0, 1, 2)), "ns_3x128": (3, 4, 3, 128, (0, 1, 2)), "pe_3x100": (2, 6, 3, 100, (0, 1)),
         "co_4x200": (2, 3, 4, 200, (0, 1))}
for name in sys.argv[1:] or list(CASES):
    d_in, d_out, L, W, gc = CASES[name]
    g = torch.Generator().manual_seed(4321)
    params = O.init_params(O.layer_sizes(d_in, L, W, d_out), "xavier", g)
    N = 160
    X = (torch.rand(N, d_in, generator=g) * 2 - 1).cuda()
    flat = O.flatten(params).cuda()
    desc = NetDesc(d_in, d_out, L, W, gc, engine=ENGINE_WIDE)
    out = {}
    for prec in (PREC_F32, PREC_BF16):
        eng = Engine(desc.with_(precision=prec))
        r = eng.forward_jet(flat, X)
        torch.cuda.synchronize()
        out[prec] = [r[0].float().cpu(), r[1].float().cpu().permute(1, 0, 2)]
    ref, got = out[PREC_F32], out[PREC_BF16]
    for qi, (a, b) in enumerate(zip(ref, got)):
        a = a.reshape(N, -1); b = b.reshape(N, -1)
        # tangents may come as (N, d_out, k) or so: report per trailing column group
        err = (a - b).abs().reshape(N // 16, 16, -1)
        scale = a.abs().max().item() + 1e-30
        per_pos = err.amax(dim=(0, 2)) / scale
        per_col = (a - b).abs().amax(dim=0) / scale
        print(f"{name} tensor{qi} shape {tuple(out[PREC_F32][qi].shape)} max rel err {per_pos.max():.2e}")
        print("   by point position in tile:", " ".join(f"{v:.0e}" for v in per_pos.tolist()))
        print("   by column:", " ".join(f"{v:.0e}" for v in per_col.tolist()))
