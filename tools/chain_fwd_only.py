"""Forward-only (loss without gradient) bf16 12x256 pass at 2^20 points: the chain forward kernel without its a_l stores
(diagnostic: with PINN_HIP_LIB pointing at a -DPINN_CHAIN_DIAG build the phase shares are printed)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd.dnn import init_flat_params
desc = NetDesc(3, 4, 12, 256, (0, 1, 2), precision=1)
spec = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), desc.grad_cols, ("h", "z", "u", "v"))
eng = Engine(desc, "cuda:0")
g = torch.Generator().manual_seed(1)
flat = init_flat_params(desc.layers, "xavier", g).cuda()
X = (torch.rand(1 << 20, 3, generator=g) * 2 - 1).cuda()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = eng.residual_loss(spec, flat, X)
    torch.cuda.synchronize(); print("forward-only pass: %.2f ms" % ((time.perf_counter() - t0) * 1e3), s.tolist())
