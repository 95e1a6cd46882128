"""forward_speed.py: the inference path (test.py:76,96 — DNN.forward over a grid; pinn_forward / pinn_forward_jet) at
2^20 and 2^22 points for the reference's network shapes: ms per call, points/s, share of the fp32 MFMA peak (plain
forward: 2 M flop per point; jet: 2 M (1 + k)) and of the HBM roof (algorithmic bytes: 4 (d_in + d_out (1 + k)) per point)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pinn_depthestimation_amd import Engine, NetDesc
from pinn_depthestimation_amd.dnn import init_flat_params
SHAPES = {"ns8x64": (3, 4, 8, 64, (0, 1, 2)), "pe10x10": (2, 6, 10, 10, (0, 1)), "co100x20": (2, 3, 100, 20, (0, 1)),
          "ns20x20": (5, 4, 20, 20, (0, 1, 2)), "ns12x256": (3, 4, 12, 256, (0, 1, 2))}
for name, (d_in, d_out, L, W, gc) in SHAPES.items():
    desc = NetDesc(d_in, d_out, L, W, gc)
    params = init_flat_params(desc.layers, "xavier", torch.Generator().manual_seed(3)).cuda()
    M = sum(a * b for a, b in zip(desc.layers[:-1], desc.layers[1:]))
    eng = Engine(desc)
    for N in (1 << 20, 1 << 22):
        if W == 256 and N > (1 << 20): continue
        X = (torch.rand(N, d_in, generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
        for what in ("forward", "forward_jet"):
            fn = (lambda: eng.forward(params, X)) if what == "forward" else (lambda: eng.forward_jet(params, X))
            for _ in range(2): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): fn()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            kk = 0 if what == "forward" else len(gc)
            flop, byts = 2.0 * M * (1 + kk) * N, 4.0 * (d_in + d_out * (1 + kk)) * N
            print(f"{name} {what} N=2^{N.bit_length() - 1}: {ms:.3f} ms  {N / ms * 1e3:.3e} points/s  "
                  f"{flop / ms / 1e9 / 157.3 * 100:.1f} % of 157.3 TF  {byts / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s")
