#!/bin/bash
# build_full_variant.sh NAME "FLAGS": rebuild every translation unit in a scratch copy with extra
# flags -> pinn_depthestimation_amd/libpinn_hip_NAME.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; FLAGS=$2
D=/tmp/pinn_variant_$NAME
rm -rf $D; mkdir -p $D/pinn_depthestimation_amd $D/include
cp $ROOT/include/*.h $D/include/
cp -r $ROOT/pinn_depthestimation_amd/csrc $D/pinn_depthestimation_amd/csrc
rm -f $D/pinn_depthestimation_amd/csrc/*.o
make -C $D/pinn_depthestimation_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $FLAGS" > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
cp $D/pinn_depthestimation_amd/libpinn_hip.so $ROOT/pinn_depthestimation_amd/libpinn_hip_$NAME.so
echo built libpinn_hip_$NAME.so
