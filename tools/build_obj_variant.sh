#!/bin/bash
# A variant of the product library with ONE translation unit recompiled with extra flags -> tools/bin/libts2d_<tag>.so (A/B runs through
# TS2D_LIBRARY_PATH + TS2D_BINDING=ctypes, tools/lease.sh `ab`; the product library is untouched).
#   usage: tools/build_obj_variant.sh <tag> <object name in triangle-splatting_amd/build, without .o> "<extra flags>"
#   e.g.   tools/build_obj_variant.sh r3d_ilp render3d_group "-mllvm -amdgpu-sched-strategy=max-ilp"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; OBJ=$2; XF=$3
python $R/triangle-splatting_amd/build.py > /dev/null
B=$R/triangle-splatting_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG -fvisibility=hidden"
case $OBJ in
  render_group_fwd) SRC=render_group; F="$F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DTSG_PART=1 -mllvm -amdgpu-sched-strategy=max-ilp";;
  render_group_bwd) SRC=render_group; F="$F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize -DTSG_PART=2";;
  render3d_group) SRC=$OBJ; F="$F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize";;
  preprocess*|shgrad|depth_normal|aux_losses|resample|optim) SRC=$OBJ; F="$F -ffp-contract=off";;
  *) SRC=$OBJ;;
esac
T=/tmp/ts2d_objvar_$TAG
mkdir -p $R/tools/bin $T
/opt/rocm/bin/hipcc $F $XF -c $R/triangle-splatting_amd/csrc/$SRC.hip -o $T/$OBJ.o
OBJS=""
for o in $B/*.o; do
  if [ "$(basename $o)" = "$OBJ.o" ]; then OBJS="$OBJS $T/$OBJ.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_$TAG.so $OBJS
echo $R/tools/bin/libts2d_$TAG.so
