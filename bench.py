#!/usr/bin/env python
"""bench.py — collocation-point residuals/sec for one full optimisation step
(forward + PDE partials + d loss/d theta + Adam) on BASELINE.json configs[1]:
3 -> 8x64 tanh -> 4, Navier_Stokes residual, 2^20 synthetic (t,x,y) points per GPU, fp32.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms run N ranks, one process per GPU.  Under torch.distributed.run the ranks already exist
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); started plainly with --gpus N > 1 this
process becomes a LAUNCHER: it never touches the GPU, checks that N devices are visible (refusing
loudly otherwise — it never reports fewer GPUs than it was asked for), starts the N ranks as child
processes on 127.0.0.1 and relays rank 0's JSON line.

A step is the arithmetic of train.py:189-193 (zero_grad, loss_func, backward, Adam.step, StepLR.step)
on the full batch, with per-iteration logging off.  Points are sharded across ranks
(weak scaling: 2^20 per GPU); ONE RCCL all-reduce of [grad | loss sums] per step (north_star:
"RCCL all-reduce of the loss gradient over xGMI").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# name: (d_in, d_out, hidden, width, grad_cols, residual, input names, output names, flop/point = 6*M*(1+k))
WORKLOADS = {
    "ns8x64": (3, 4, 8, 64, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v"), 695_424),          # BASELINE configs[1]
    "pe8x64": (2, 6, 8, 64, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"), 523_776),  # configs[2], 8x64
    "pe10x10": (2, 6, 10, 10, (0, 1), "physics_equation", ("x", "y"), ("h", "U", "V", "eta_mean", "Hrms", "k"), 17_400),  # configs[2] as written
    "ns12x256": (3, 4, 12, 256, (0, 1, 2), "Navier_Stokes", ("t", "x", "y"), ("h", "z", "u", "v"), 17_330_688),  # configs[3] shape
    "co100x20": (2, 3, 100, 20, (0, 1), "continuity_ftemp", ("x", "y"), ("U", "V", "h"), 6 * (2 * 20 + 99 * 400 + 60) * 3),  # config_CMB_h.json net
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.8    # same guide: dense bf16 MFMA = 16x the fp32 matrix rate (~2.5 PF)
PTS_PER_GPU = 1 << 20
# HBM bytes per point from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE x2 on
# gfx950 + WRITE_SIZE), keyed by (workload, bf16)
PMC_SUMMARIES = {
    ("ns8x64", False): ("profiles/r03/fused_r03_pmc_summary.json", "hbm_bytes_per_point"),
    ("pe10x10", False): ("profiles/r03/pe10x10_batch_pmc_summary.json", "hbm_bytes_per_point"),
    ("co100x20", False): ("profiles/r03/co100x20_batch_pmc_summary.json", "hbm_bytes_per_point"),
    ("ns12x256", True): ("profiles/r02/wide_bf16_pmc_summary.json", "hbm_bytes_per_point"),
    ("ns12x256", False): ("profiles/r02/wide_f32_pmc_summary.json", "hbm_bytes_per_point"),
}


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
    import torch
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


def run_lbfgs_stage(args):
    """BASELINE configs[4] ("full-batch N = 1 M L-BFGS, report closure evals/s x N"): 3->8x64->4 Navier-Stokes on
    args.points points, 50 Adam steps as warm start (train.py:188-193), then the L-BFGS stage two ways — the live one,
    ONE torch.optim.LBFGS.step(closure) (train.py:116-125,195-200; lbfgs.FlatLBFGS), and the stale-bytecode one,
    SciPy L-BFGS-B over a flat float64 vector (lbfgsb.LBFGSBOptimizer) — for args.steps iterations each.  Reported:
    closure evaluations/s x N and the share of the wall time spent outside the loss+gradient call (optimizer math,
    float(loss) synchronisations, the P-float host copies of the SciPy driver)."""
    import torch
    from pinn_depthestimation_amd.trainer import PINN
    from pinn_depthestimation_amd.lbfgsb import LBFGSBOptimizer
    N, iters = args.points, max(args.steps, 1)
    cfg = {"layers": {"input_features": 3, "hidden_layers": 8, "hidden_width": 64, "output_features": 4},
           "adam_optimizer": {"max_it": 50, "learning_rate": 1e-4, "scheduler_step_size": 10000, "scheduler_gamma": 0.8},
           "lbfgs_optimizer": {"max_it": iters, "learning_rate": 1, "max_evaluation": None, "history_size": 100,
                               "tolerance_grad": 0.0, "tolerance_change": 0.0, "line_search_fn": "strong_wolfe"},
           "loss": {"weight_fid_loss": 1, "weight_res_loss": 1},
           "data_fidelity": {"inputs": ["t", "x", "y"], "outputs": []},
           "data_residual": {"inputs": {k: {"requires_grad": ["true"]} for k in "txy"}, "outputs": ["h", "z", "u", "v"]}}
    X = (torch.rand(N, 3, generator=torch.Generator().manual_seed(1234)) * 2 - 1).numpy()
    out = {}
    # Each driver runs the whole stage TWICE from the same start; the second run is the one reported.  The first pays
    # the process's one-time costs (library handles of the fp64 triangular solves, code objects of the L-BFGS
    # kernels, history allocation, SciPy's import): ~0.5 s that belong to the process, not to the stage
    # (tools/lbfgs_stage_profile.py: 593 ms cold, 120 ms warm for the same 17 evaluations).
    for impl in ("torch.optim.LBFGS", "torch.optim.LBFGS", "scipy L-BFGS-B", "scipy L-BFGS-B"):
        torch.manual_seed(1234)
        tr = PINN(None, None, X, cfg, log_every=1000, checkpoint_every=0, engine=args.engine)
        tr.train_adam(50)
        for _ in range(3):
            tr.closure()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.closure()
        torch.cuda.synchronize()
        t_closure = (time.perf_counter() - t0) / 10          # the loss + gradient call alone, back to back
        e0 = tr.iter
        t0 = time.perf_counter()
        if impl == "torch.optim.LBFGS":
            tr.optimizer_LBFGS.step(tr.closure)
        else:
            LBFGSBOptimizer(tr, {"maxiter": iters, "maxfun": 10 * iters, "ftol": 0.0, "gtol": 0.0}).minimize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        evals = tr.iter - e0
        out[impl] = {"closure_evals": evals, "seconds": dt, "evals_per_s": evals / dt, "points_per_s": evals * N / dt,
                     "closure_ms": t_closure * 1e3, "share_outside_closure": max(0.0, 1.0 - evals * t_closure / dt),
                     "final_loss": float(tr.last[2])}
        log(f"{impl}: {evals} closure evaluations in {dt:.3f} s")
    live = out["torch.optim.LBFGS"]
    print(json.dumps({
        "metric": "L-BFGS closure evaluations/s x N (full-batch residual + gradient inside the L-BFGS stage)",
        "value": live["points_per_s"], "unit": "residual-points/s", "n_gpus": 1, "steps": iters, "warmup": 50,
        "ms_per_step": live["seconds"] / max(live["closure_evals"], 1) * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4]: L-BFGS fine-tune stage after 50 Adam steps, 3->8x64 tanh->4 MLP, "
                               f"Navier_Stokes residual, {N} synthetic (t,x,y) points, full batch; ms_per_step is per closure evaluation",
                   "points_per_gpu": N, "lbfgs_iterations": iters, "drivers": out}}), flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=PTS_PER_GPU, help="points per GPU")
    ap.add_argument("--engine", type=int, default=0, help="0 auto, 1 generic, 2 fused, 3 wide, 4/5 fused tile/coop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="bf16 MFMA operands (wide engine only; extra evidence)")
    ap.add_argument("--workload", default="ns8x64", choices=sorted(WORKLOADS) + ["lbfgs8x64"],
                    help="default = the headline BASELINE configs[1]; others are extra evidence, not the contract line")
    ap.add_argument("--test-evaluator", default=os.environ.get("PINN_BENCH_TEST_EVALUATOR"),
                    help="module:factory of a CPU evaluator (tests only: rehearses the N-rank launcher and the "
                         "all-reduce path with gloo on a machine without GPUs; the line it prints is marked invalid)")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` with no torch.distributed.run around it ------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count() -> int:
    """GPUs this process could open, counted WITHOUT loading HIP (no torch.cuda, no libamdhip64): the launcher must
    stay a process that never touched the GPU.  An explicit *_VISIBLE_DEVICES list wins; otherwise the KFD topology
    nodes that have SIMDs (CPUs are nodes too) capped by the DRM render nodes this container may actually open; -1 when
    sysfs has no KFD topology to read (the launcher then leaves the check to the ranks, which use the real runtime)."""
    import glob
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    kfd = 0
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        return -1              # sysfs is not telling (no KFD topology in this container): the ranks find out for themselves
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(props):
                k, _, val = line.partition(" ")
                if k == "simd_count" and int(val) > 0:
                    kfd += 1
        except (OSError, ValueError):
            pass
    render = [d for d in glob.glob("/dev/dri/renderD*") if os.access(d, os.R_OK | os.W_OK)]
    return min(kfd, len(render)) if render else 0


def launch_ranks(args, argv) -> int:
    """Start args.gpus ranks of this script as child processes and wait for them.  This process makes NO GPU call and
    loads no GPU runtime: the device count comes from sysfs (visible_gpu_count); every child checks again for itself
    with the real runtime and refuses to share a card."""
    n = args.gpus
    if not args.test_evaluator:
        have = visible_gpu_count()
        if 0 <= have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; refusing to run a smaller job under "
                  f"that name", file=sys.stderr, flush=True)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PINN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    log(f"launcher: started {n} ranks on 127.0.0.1:{port} (pids {[p.pid for p in procs]})")
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    log(f"launcher: rank pid {p.pid} exited with {code}; stopping the others")
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.workload == "lbfgs8x64":
        if args.gpus != 1:
            print("bench.py: --workload lbfgs8x64 is a one-GPU measurement", file=sys.stderr, flush=True)
            sys.exit(2)
        return run_lbfgs_stage(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: the job would not be the one named; refusing",
              file=sys.stderr, flush=True)
        sys.exit(2)
    cpu_test = bool(args.test_evaluator)
    if cpu_test:
        dev = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} wants cuda:{local} but {torch.cuda.device_count()} GPU(s) are visible",
                  file=sys.stderr, flush=True)
            sys.exit(2)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    force_dist = os.environ.get("PINN_BENCH_FORCE_DIST") == "1"     # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        if cpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    d_in, d_out, hidden, width, gcols, res_name, in_names, out_names, flop_pt = WORKLOADS[args.workload]
    from pinn_depthestimation_amd import NetDesc, ResidualSpec
    from pinn_depthestimation_amd.dnn import init_flat_params
    desc = NetDesc(d_in, d_out, hidden, width, gcols, engine=args.engine, precision=1 if args.bf16 else 0)
    spec = ResidualSpec.from_names(res_name, in_names, desc.grad_cols, out_names)
    if cpu_test:
        import importlib
        mod, fn = args.test_evaluator.split(":")
        eng = getattr(importlib.import_module(mod), fn)(desc, spec)
    else:
        from pinn_depthestimation_amd import Engine
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

    def mk_events():
        return None if cpu_test else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev = [mk_events() for _ in range(args.steps)]         # around the loss+gradient kernel chain
    ev_ar = [mk_events() for _ in range(args.steps)]      # around the all-reduce
    host_ar = [0.0]

    def step(i, timed_idx=None):
        grad.zero_()
        timed = timed_idx is not None and not cpu_test
        if timed: ev[timed_idx][0].record()
        eng.residual_loss_grad(spec, scale, params, X, grad, sums=sums)
        if timed: ev[timed_idx][1].record()
        if dist is not None:
            if timed: ev_ar[timed_idx][0].record()
            t_ar = time.perf_counter()
            dist.all_reduce(buf)
            if timed: ev_ar[timed_idx][1].record()
            elif timed_idx is not None: host_ar[0] += time.perf_counter() - t_ar
        lr = lr0 * gamma ** (i // sched_step)
        eng.adam_step(params, grad, m, v, i + 1, lr)

    def barrier():
        if not cpu_test: torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not cpu_test: torch.cuda.synchronize()

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
        if cpu_test:
            kern_ms = None
            ar_ms = host_ar[0] / args.steps * 1e3 if dist is not None else 0.0
        else:
            kern_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
            ar_ms = sum(a.elapsed_time(b) for a, b in ev_ar) / args.steps if dist is not None else 0.0
        traffic, traffic_src = None, None   # HBM bytes per launch from the committed PMC passes of this same command
        src = PMC_SUMMARIES.get((args.workload, bool(args.bf16)))
        if src is not None and not cpu_test and args.engine in (0, 2, 3):
            try:
                pm = json.load(open(os.path.join(ROOT, src[0])))
                traffic = pm[src[1]] * N
                traffic_src = f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, {src[0]}"
            except Exception:
                pass
        achieved = N * flop_pt / (kern_ms * 1e-3) / 1e12 if kern_ms else None
        peak = PEAK_BF16_MFMA_TFLOPS if args.bf16 else PEAK_F32_MFMA_TFLOPS
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "collocation-point residuals/sec (fwd+PDE-grad+Adam)",
            "value": n_global * args.steps / dt, "unit": "residual-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.bf16 else "f32",
            "data": "synthetic" if not cpu_test else "INVALID: test evaluator on CPU (launcher rehearsal, not a measurement)",
            "config": {"workload": (f"BASELINE configs[1]: " if args.workload == "ns8x64" else f"[{args.workload}] ") +
                                   f"{d_in}->{hidden}x{width} tanh->{d_out} MLP, {res_name} residual, "
                                   f"{N} synthetic ({','.join(in_names)}) points per GPU, full-batch Adam step",
                       "points_per_gpu": N, "global_points": n_global, "params": P,
                       "parallelism": f"dp{world}", "engine": args.engine, "final_loss": loss,
                       "allreduce_bytes": (P + nt) * 4 if dist is not None else 0,
                       "allreduce_ms": ar_ms, "allreduce_frac_of_step": ar_ms / ms_step if ms_step > 0 else 0.0},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": "pinn_residual_loss_grad (fwd jet + residual + reverse sweep)",
                         "kernel_ms": kern_ms, "flop_per_point": flop_pt},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "ns8x64" and not cpu_test:
            out["cpu_baseline"] = cpu_baseline(usable_cpus())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
