#!/bin/bash
# build_variant.sh NAME "EXTRA FLAGS": an experimental copy of the library at pinn_depthestimation_amd/libpinn_NAME.so
# (git-ignored; travels with gpurun; select it with PINN_HIP_LIB).
R="$(cd "$(dirname "$0")/.." && pwd)"
D=/tmp/variant_$1; rm -rf $D; mkdir -p $D; cp -r $R/pinn_depthestimation_amd/csrc $D/; rm -f $D/csrc/*.o; mkdir -p $D/include; cp $R/include/pinn_hip.h $D/include/
sed -i 's#\.\./\.\./include/pinn_hip.h#../include/pinn_hip.h#' $D/csrc/common.h $D/csrc/Makefile
make -C $D/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $2" OUT=$R/pinn_depthestimation_amd/libpinn_$1.so > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
ls -la $R/pinn_depthestimation_amd/libpinn_$1.so
