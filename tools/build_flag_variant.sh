#!/bin/bash
# A variant of the WHOLE product library compiled with extra flags (every source, its own object directory) -> tools/bin/libts2d_<tag>.so.
# The product library and its objects are untouched; tests / bench pick the variant through TS2D_LIBRARY_PATH.
#   usage: tools/build_flag_variant.sh <tag> <extra flags...>      e.g.  tools/build_flag_variant.sh qmask -DTS2D_QMASK
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
O=/tmp/ts2d_flagvar_$TAG
mkdir -p $O $R/tools/bin
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG -fvisibility=hidden"
pids=""
for SRC in preprocess preprocess3d shgrad photometric depth_normal aux_losses resample knn model_update optim binning select render3d_group api; do
  X=""
  case $SRC in
    render*) X="-mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize";;
    preprocess*|shgrad|depth_normal|aux_losses|resample|optim) X="-ffp-contract=off";;
  esac
  /opt/rocm/bin/hipcc $F $X "$@" -c $R/triangle-splatting_amd/csrc/$SRC.hip -o $O/$SRC.o &
  pids="$pids $!"
done
# the 2D blend kernels: one source, two translation units (build.py)
BX="-mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F $BX -DTSG_PART=1 -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c $R/triangle-splatting_amd/csrc/render_group.hip -o $O/render_group_fwd.o &
pids="$pids $!"
/opt/rocm/bin/hipcc $F $BX -DTSG_PART=2 "$@" -c $R/triangle-splatting_amd/csrc/render_group.hip -o $O/render_group_bwd.o &
pids="$pids $!"
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_$TAG.so $O/*.o
echo $R/tools/bin/libts2d_$TAG.so
# the matching lab library (state reader, test hooks, the measurement kernels of earlier rounds): the variant's objects, api.hip compiled with -DTS2D_LAB
mkdir -p $O/lab
pids=""
for SRC in render render3d render_q8 lab_hooks api; do
  X=""
  case $SRC in
    render*) X="-mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize";;
    api) X="-DTS2D_LAB";;
  esac
  D=$R/tools/lab; [ $SRC = api ] && D=$R/triangle-splatting_amd/csrc   # the lab kernels live in tools/lab/ since round 6
  /opt/rocm/bin/hipcc $F $X -I$R/triangle-splatting_amd/csrc "$@" -c $D/$SRC.hip -o $O/lab/$SRC.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
OBJS=$(ls $O/*.o | grep -v "/api.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_lab_$TAG.so $OBJS $O/lab/*.o
echo $R/tools/bin/libts2d_lab_$TAG.so
