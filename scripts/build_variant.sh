#!/bin/bash
# build/variants/<name>.so: the library with extra compiler flags on vspg_capi.hip (the wavefront pipeline's objects are reused unless
# WF=1), for same-box A/Bs through VSPG_LIB (scripts/gpu_variants.sh):  bash scripts/build_variant.sh <name> "<flags>"
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../vspg-pbrt-v4_amd/csrc"
HIPFLAGS=$(make -pn 2>/dev/null | sed -n "s/^HIPFLAGS = //p" | head -1 | sed "s/\$(ARCH)/gfx950/")
OUT=../../build/variants; mkdir -p $OUT
/opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -c -o $OUT/$NAME.capi.o vspg_capi.hip
OBJS="vspg_wf_grid.o vspg_wf_nvdb.o vspg_fast1.o vspg_fast2.o"
if [ "${WF:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -c -o $OUT/$NAME.wf_grid.o vspg_wf_grid.hip &
  /opt/rocm/bin/hipcc $HIPFLAGS $FLAGS -c -o $OUT/$NAME.wf_nvdb.o vspg_wf_nvdb.hip & wait
  OBJS="$OUT/$NAME.wf_grid.o $OUT/$NAME.wf_nvdb.o vspg_fast1.o vspg_fast2.o"
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT/$NAME.so $OUT/$NAME.capi.o $OBJS
echo built $OUT/$NAME.so
