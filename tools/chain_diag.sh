#!/bin/bash
# chain_diag.sh: build a DIAGNOSTIC copy of the library with per-phase s_memtime stamps in the chain kernels
# (-DPINN_CHAIN_DIAG) into /tmp and run one bf16 bench step with it: prints each kernel's phase shares.
R="$(cd "$(dirname "$0")/.." && pwd)"
D=/tmp/chain_diag_build; rm -rf $D; mkdir -p $D; cp -r $R/pinn_depthestimation_amd/csrc $D/; mkdir -p $D/include; cp $R/include/pinn_hip.h $D/include/
sed -i 's#\.\./\.\./include/pinn_hip.h#../include/pinn_hip.h#' $D/csrc/common.h $D/csrc/Makefile
make -C $D/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -DPINN_CHAIN_DIAG $CHAIN_DIAG_EXTRA" OUT=$D/libpinn_diag.so > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
if [ "$1" = "fwdonly" ]; then PINN_HIP_LIB=$D/libpinn_diag.so python3 $R/tools/chain_fwd_only.py 2>&1 | tail -6; else PINN_HIP_LIB=$D/libpinn_diag.so python3 $R/bench.py --workload ns12x256 --bf16 --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | grep -E "CHAIN_DIAG|ms_per_step" | cut -c1-400 | tail -10; fi
