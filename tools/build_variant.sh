#!/bin/bash
# A variant of the product library with ONE source recompiled with extra flags -> tools/bin/libts2d_<tag>.so (for A/B runs through
# TS2D_LIBRARY_PATH; the product library is untouched).   usage: tools/build_variant.sh <tag> <source.hip> <extra flags...>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; SRC=$2; shift 2
python $R/triangle-splatting_amd/build.py > /dev/null
B=$R/triangle-splatting_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG -fvisibility=hidden"
case $SRC in
  render*) F="$F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize";;
  preprocess*|shgrad*|depth_normal*|optim*) F="$F -ffp-contract=off";;
esac
mkdir -p $R/tools/bin /tmp/ts2d_var_$TAG
N=$(basename $SRC .hip)
/opt/rocm/bin/hipcc $F "$@" -c $R/triangle-splatting_amd/csrc/$SRC -o /tmp/ts2d_var_$TAG/$N.o
OBJS=""
for o in $B/*.o; do
  if [ "$(basename $o)" = "$N.o" ]; then OBJS="$OBJS /tmp/ts2d_var_$TAG/$N.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_$TAG.so $OBJS
echo $R/tools/bin/libts2d_$TAG.so
