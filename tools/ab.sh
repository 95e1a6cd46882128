#!/bin/bash
# ab.sh "label:lib:pairflag" ... : interleaved A/B of bench.py over variant libraries on one GPU box
# (device-to-device variance is 2-15 %, so variants are only comparable within one call)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for spec in "$@"; do
    IFS=: read label lib pair <<< "$spec"
    ms=$(PINN_HIP_LIB=$PWD/pinn_depthestimation_amd/$lib PINN_FUSED_PAIR=$pair timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $AB_ARGS 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "$label rep$rep $ms"
  done
done
