#!/bin/bash
# build_tu_variant.sh NAME "EXTRA FLAGS" TU [TU ...]: a copy of the library in which the named translation units (e.g.
# pinn_fused_w64) are compiled with the Makefile's own flags for them PLUS the extra flags; everything else is linked
# from the in-tree objects (run `make` first).  Output: pinn_depthestimation_amd/libpinn_NAME.so (git-ignored; select it
# with PINN_HIP_LIB).  The A/B tool behind the per-TU code-generation switches in the Makefile.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; C=$R/pinn_depthestimation_amd/csrc
NAME=$1; EXTRA=$2; shift 2
D=/tmp/tuvar_$NAME; rm -rf $D; mkdir -p $D
EXCL=""
for TU in "$@"; do
  FLAGS=$(make -C $C -n -B $TU.o 2>/dev/null | grep -- "-c $TU.hip" | sed -E "s#^.*hipcc (.*) -c $TU.hip.*#\1#")
  (cd $C && /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $TU.hip -o $D/$TU.o) &
  EXCL="$EXCL|^$TU.o\$"
done
wait
OBJS=$(cd $C && ls *.o | grep -Ev "${EXCL#|}")
(cd $C && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $D/*.o -o $R/pinn_depthestimation_amd/libpinn_$NAME.so)
ls -la $R/pinn_depthestimation_amd/libpinn_$NAME.so
