#!/bin/bash
# build_variant.sh NAME SRC "FLAGS": rebuild one translation unit with extra -D flags and link a
# variant library pinn_depthestimation_amd/libpinn_hip_NAME.so (A/B runs: PINN_HIP_LIB=<that path>)
set -e
cd "$(dirname "$0")/../pinn_depthestimation_amd/csrc"
NAME=$1; SRC=$2; FLAGS=$3
OBJ=/tmp/variant_${NAME}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $FLAGS -c $SRC -o $OBJ
OTHERS=$(ls *.o | grep -v "^$(basename $SRC .hip).o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS $OBJ -o ../libpinn_hip_${NAME}.so
echo built ../libpinn_hip_${NAME}.so
