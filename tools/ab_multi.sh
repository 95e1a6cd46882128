#!/bin/bash
# ab_multi.sh "workload args" lib1 lib2 ... — same-box A/B of variant libraries on one bench.py workload
cd "$(dirname "$0")/.."
W=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "$W $lib $(PINN_HIP_LIB=$PWD/pinn_depthestimation_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline $W 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  done
done
