#!/bin/bash
# A variant of the product library with SEVERAL sources recompiled with extra flags -> tools/bin/libts2d_<tag>.so.
#   usage: tools/build_multi_variant.sh <tag> src1.hip "<flags1>" [src2.hip "<flags2>" ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
python $R/triangle-splatting_amd/build.py > /dev/null
B=$R/triangle-splatting_amd/build
mkdir -p $R/tools/bin /tmp/ts2d_var_$TAG
REPL=""
while [ $# -ge 2 ]; do
  SRC=$1; FL=$2; shift 2
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG -fvisibility=hidden"
  case $SRC in
    render*) F="$F -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize";;
    preprocess*|shgrad*|depth_normal*|optim*) F="$F -ffp-contract=off";;
  esac
  N=$(basename $SRC .hip)
  /opt/rocm/bin/hipcc $F $FL -c $R/triangle-splatting_amd/csrc/$SRC -o /tmp/ts2d_var_$TAG/$N.o &
  REPL="$REPL $N.o"
done
wait
OBJS=""
for o in $B/*.o; do
  n=$(basename $o)
  case " $REPL " in *" $n "*) OBJS="$OBJS /tmp/ts2d_var_$TAG/$n";; *) OBJS="$OBJS $o";; esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_$TAG.so $OBJS
echo $R/tools/bin/libts2d_$TAG.so
