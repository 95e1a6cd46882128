"""Times the engine's entry points separately (HIP events) to see where a step goes.
usage: python tools/bench_phases.py [points]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
from pinn_depthestimation_amd.dnn import init_flat_params

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
desc = NetDesc(3, 4, 8, 64, (0, 1, 2))
spec = ResidualSpec.from_names("Navier_Stokes", ("t", "x", "y"), desc.grad_cols, ("h", "z", "u", "v"))
eng = Engine(desc, "cuda")
g = torch.Generator().manual_seed(1)
params = init_flat_params(desc.layers, "xavier", g).cuda()
X = (torch.rand(N, 3, generator=g) * 2 - 1).cuda()
scale = torch.full((3,), 1.0 / N, device="cuda")
grad = torch.zeros(desc.n_params, device="cuda")
T = torch.rand(N, 4, generator=g).cuda()
cs = torch.full((4,), 1.0 / N, device="cuda")

def timeit(name, fn, flop_per_pt, reps=int(os.environ.get("REPS", "5"))):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:28s} {ms:8.3f} ms  {N/ms/1e3:8.2f} Mpts/s  {N*flop_per_pt/ms/1e9:7.2f} TFLOP/s ({N*flop_per_pt/ms/1e9/157.3*100:4.1f}% of fp32 MFMA)", flush=True)

M = 29120
timeit("forward (k=0)", lambda: eng.forward(params, X), 2 * M)
timeit("forward_jet (k=3)", lambda: eng.forward_jet(params, X), 2 * M * 4)
timeit("residual_loss (no grad)", lambda: eng.residual_loss(spec, params, X), 2 * M * 4)
timeit("residual_loss_grad", lambda: eng.residual_loss_grad(spec, scale, params, X, grad), 6 * M * 4)
timeit("mse_loss_grad (k=0)", lambda: eng.mse_loss_grad(params, X, T, [0, 1, 2, 3], cs, grad), 6 * M)
