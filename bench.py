#!/usr/bin/env python
"""bench.py — collocation-point residuals/sec for one full optimisation step
(forward + PDE partials + d loss/d theta + Adam) on BASELINE.json configs[1]:
3 -> 8x64 tanh -> 4, Navier_Stokes residual, 2^20 synthetic (t,x,y) points per GPU, fp32.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is the arithmetic of train.py:189-193 (zero_grad, loss_func, backward,
Adam.step, StepLR.step) on the full batch, with per-iteration logging off.
Points are sharded across ranks (weak scaling: 2^20 per GPU); one RCCL all-reduce of
[grad | loss sums] per step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_POINT = 695_424          # 6*M*(1+k), M = 29 120, k = 3 (3->8x64->4)

# name: (d_in, d_out, hidden, width, grad_cols, residual, input names, output names, flop/point = 6*M*(1+k))
WORKLOADS = {
    "ns8x64": (3, 4, 8, 64, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v"), 695_424),          # BASELINE configs[1]
    "pe8x64": (2, 6, 8, 64, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"), 523_776),  # configs[2], 8x64
    "pe10x10": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"), 17_400),  # configs[2] as written
    "ns12x256": (3, 4, 12, 256, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v"), 17_330_688),  # configs[3] shape, fp32
    "co100x20": (2, 3, 100, 20, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h"), 6 * (2 * 20 + 99 * 400 + 60) * 3),  # config_CMB_h.json net
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.8    # same guide: dense bf16 MFMA = 16x the fp32 matrix rate (~2.5 PF)
PTS_PER_GPU = 1 << 20


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 32))


def cpu_baseline(threads: int):
    """Oracle (autograd formulation of the reference, torch CPU) timed on a bounded sample:
    N = 10 000 points (BASELINE configs[0] size), 1 warm-up + steps until ~12 s."""
    from oracle import pinn_oracle as O
    torch.set_num_threads(threads)
    log(f"cpu baseline on {threads} threads")
    g = torch.Generator().manual_seed(1234)
    layers = O.layer_sizes(3, 8, 64, 4)
    params = [p.requires_grad_(True) for p in O.init_params(layers, "xavier", g)]
    N = 10_000
    X = torch.rand(N, 3, generator=g) * 2 - 1
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad()
        loss = O.residual_loss(params, X, "Navier_Stokes", [0, 1, 2], [0, 1, 2, 3], (0, 1, 2))
        loss.backward()
        opt.step()

    step()
    t0, n = time.perf_counter(), 0
    while n < 1 or (time.perf_counter() - t0 < 12.0 and n < 200):
        step(); n += 1
    dt = time.perf_counter() - t0
    log(f"cpu baseline: {n} steps in {dt:.1f} s")
    return {"value": N * n / dt, "unit": "residual-points/s", "cores": threads, "kind": "port",
            "sample": f"oracle/pinn_oracle.py (13x autograd.grad + double backward + torch Adam), "
                      f"N={N} points x {n} steps, 3->8x64->4 Navier_Stokes, fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=PTS_PER_GPU, help="points per GPU")
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 generic, 2 fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="bf16 MFMA operands (wide engine only; extra evidence)")
    ap.add_argument("--workload", default="ns8x64", choices=sorted(WORKLOADS),
                    help="default = the headline BASELINE configs[1]; others are extra evidence, not the contract line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("PINN_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    from pinn_depthestimation_amd import Engine, NetDesc, ResidualSpec
    from pinn_depthestimation_amd.dnn import init_flat_params

    d_in, d_out, hidden, width, gcols, res_name, in_names, out_names, flop_pt = WORKLOADS[args.workload]
    desc = NetDesc(d_in, d_out, hidden, width, gcols, engine=args.engine, precision=1 if args.bf16 else 0)
    spec = ResidualSpec.from_names(res_name, in_names, desc.grad_cols, out_names)
    eng = Engine(desc, dev)
    P = desc.n_params
    g = torch.Generator().manual_seed(1234)              # same weights on every rank
    params = init_flat_params(desc.layers, "xavier", g).to(dev)
    gx = torch.Generator().manual_seed(1234 + 7919 * rank)  # each rank its own shard of points
    N = args.points
    if res_name == "physics_equation":                    # keep eta_mean + h away from 0
        off_b = P - d_out
        params[off_b + out_names.index("h")] = 0.75
        params[off_b + out_names.index("eta_mean")] = 0.0
    X = (torch.rand(N, d_in, generator=gx) * 2 - 1).to(dev)
    n_global = N * world
    nt = spec.n_terms
    scale = torch.full((nt,), 1.0 / n_global, device=dev)
    buf = torch.zeros(P + nt, device=dev)                 # [grad | term sums]: ONE all-reduce per step
    grad, sums = buf[:P], buf[P:]
    m, v = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
    lr0, gamma, sched_step = 1e-4, 0.8, 10000            # config_CMB.json:11-16
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(i, timed_idx=None):
        grad.zero_()
        if timed_idx is not None: ev[timed_idx][0].record()
        eng.residual_loss_grad(spec, scale, params, X, grad, sums=sums)
        if timed_idx is not None: ev[timed_idx][1].record()
        if dist is not None:
            dist.all_reduce(buf)
        lr = lr0 * gamma ** (i // sched_step)
        eng.adam_step(params, grad, m, v, i + 1, lr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: {N} points, P={P}, engine={args.engine}; warm-up")
    for i in range(args.warmup):
        step(i)
    barrier()
    log("timing")
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, i)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    loss = float((sums * scale).sum())
    log(f"{args.steps} steps in {dt:.3f} s; loss {loss:.5e}")

    if rank == 0:
        kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
        traffic = None   # HBM bytes per launch from the committed PMC passes of this same command
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01", "fused_v10_pmc_summary.json")))
            if args.engine in (0, 2) and args.workload == "ns8x64":
                traffic = pm["hbm_bytes_per_point"] * N
        except Exception:
            pass
        achieved = N * flop_pt / (kern_ms * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if args.bf16 else PEAK_F32_MFMA_TFLOPS
        out = {
            "metric": "collocation-point residuals/sec (fwd+PDE-grad+Adam)",
            "value": n_global * args.steps / dt, "unit": "residual-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.bf16 else "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[1]: " if args.workload == "ns8x64" else f"[{args.workload}] ") +
                                   f"{d_in}->{hidden}x{width} tanh->{d_out} MLP, {res_name} residual, "
                                   f"{N} synthetic ({','.join(in_names)}) points per GPU, full-batch Adam step",
                       "points_per_gpu": N, "global_points": n_global, "params": P,
                       "parallelism": f"dp{world}", "engine": args.engine, "final_loss": loss},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                         "traffic_source": ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, "
                                            "profiles/r01/fused_v10_pmc_summary.json") if traffic is not None else None,
                         "kernel": "pinn_residual_loss_grad (fwd jet + residual + reverse sweep)",
                         "kernel_ms": kern_ms, "flop_per_point": flop_pt},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "ns8x64":
            out["cpu_baseline"] = cpu_baseline(usable_cpus())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
