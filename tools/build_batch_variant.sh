#!/bin/bash
# build_batch_variant.sh NAME "EXTRA FLAGS": like build_variant.sh, but only the batch-kernel translation units and
# pinn_fused.hip are recompiled with the extra flags (the other objects are taken from the main build): ~1 min.
R="$(cd "$(dirname "$0")/.." && pwd)"
D=/tmp/variant_$1; rm -rf $D; mkdir -p $D; cp -r $R/pinn_depthestimation_amd/csrc $D/; rm -f $D/csrc/pinn_fused_batch*.o $D/csrc/pinn_fused.o; mkdir -p $D/include; cp $R/include/pinn_hip.h $D/include/
sed -i 's#\.\./\.\./include/pinn_hip.h#../include/pinn_hip.h#' $D/csrc/common.h $D/csrc/Makefile
touch -d '2000-01-01' $D/csrc/*.h $D/csrc/*.inc $D/csrc/*.hip $D/include/*.h
make -C $D/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc $2" OUT=$R/pinn_depthestimation_amd/libpinn_$1.so > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
ls -la $R/pinn_depthestimation_amd/libpinn_$1.so
