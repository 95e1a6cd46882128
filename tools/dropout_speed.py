"""dropout_speed.py: loss + gradient of the headline shape with nn.Dropout(p = 0.1) in training mode, fused tile kernel
(k_fused<..., DROP>) against the generic engine's kernels, and against p = 0."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd.dnn import init_flat_params
from pinn_depthestimation_amd._lib import ENGINE_FUSED, ENGINE_GENERIC
N = 1 << 20
X = (torch.rand(N, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
for tag, p, e, n in (("p=0 fused", 0.0, ENGINE_FUSED, N), ("p=0.1 fused", 0.1, ENGINE_FUSED, N), ("p=0.1 generic", 0.1, ENGINE_GENERIC, N // 8)):
    desc = NetDesc(3, 4, 8, 64, (0, 1, 2), dropout_p=p)
    spec = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), desc.grad_cols, ("h", "z", "u", "v"))
    params = init_flat_params(desc.layers, "xavier", torch.Generator().manual_seed(3)).cuda()
    eng = Engine(desc); eng.dropout_seed = 7
    grad = torch.zeros(desc.n_params, device="cuda"); scale = torch.full((3,), 1.0 / n, device="cuda")
    Xn = X[:n].contiguous()
    for _ in range(2): eng.residual_loss_grad(spec, scale, params, Xn, grad, engine=e)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): eng.residual_loss_grad(spec, scale, params, Xn, grad, engine=e)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"{tag}: {ms:.3f} ms per {n} points = {n / ms * 1e3:.3e} points/s")
