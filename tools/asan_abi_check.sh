#!/bin/bash
# asan_abi_check.sh: build libpinn_hip.so with AddressSanitizer + UBSan on the HOST side (device code unchanged) and run
# the CPU tests that drive the C-ABI's argument validation, engine selection, workspace queries and error reporting
# through it (no GPU: every call returns before a launch).  SURVEY §5 "race detection / sanitizers".
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
make -C $R/pinn_depthestimation_amd/csrc asan > /tmp/pinn_asan_build.log 2>&1 || { tail -20 /tmp/pinn_asan_build.log; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
cd $R
PINN_HIP_LIB=/tmp/pinn_asan/libpinn_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
python -m pytest tests/test_host_cpu.py tests/test_abi_validation_cpu.py -q -p no:cacheprovider \
  -k "exports or param_count or dropout or abi or refused or validation" "$@"
