#!/bin/bash
# ab_multi.sh — same-box A/B of variant libraries on the non-headline workloads (edit the list below)
cd "$(dirname "$0")/.."
run() { PINN_HIP_LIB=$PWD/pinn_depthestimation_amd/$1 timeout -k 10 200 python bench.py --no-cpu-baseline $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
  for lib in libpinn_hip.so libpinn_hip_wf_ilp.so libpinn_hip_wf_mem.so; do echo "ns12x256 f32 $lib $(run $lib '--workload ns12x256 --steps 3 --warmup 1')"; done
  for lib in libpinn_hip.so libpinn_hip_wb_ilp.so libpinn_hip_wb_mem.so; do echo "ns12x256 bf16 $lib $(run $lib '--workload ns12x256 --bf16 --steps 3 --warmup 1')"; done
  for lib in libpinn_hip.so libpinn_hip_co_ilp.so libpinn_hip_co_mem.so; do echo "coop243 $lib $(PINN_HIP_LIB=$PWD/pinn_depthestimation_amd/$lib REPS=1000 timeout -k 10 100 python tools/bench_phases.py 243 2>&1 | grep residual_loss_grad | cut -c1-50)"; done
done
