#!/bin/bash
# A variant of the product library whose 2D blend kernels (csrc/render_group.hip, two translation units since round 6) are recompiled with
# other flags -> tools/bin/libts2d_<tag>.so (for A/B runs through TS2D_LIBRARY_PATH; the product library is untouched).
#   usage: tools/build_blend_variant.sh <tag> "<flags of the forward unit>" "<flags of the backward unit>"
#   e.g.   tools/build_blend_variant.sh nocarry "-DTSG_CARRY=0 -mllvm -amdgpu-sched-strategy=max-ilp" "-DTSG_CARRY=0"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; FF=$2; BF=$3
python $R/triangle-splatting_amd/build.py > /dev/null
B=$R/triangle-splatting_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG -fvisibility=hidden -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
T=/tmp/ts2d_blendvar_$TAG
mkdir -p $R/tools/bin $T
/opt/rocm/bin/hipcc $F -DTSG_PART=1 $FF -c $R/triangle-splatting_amd/csrc/render_group.hip -o $T/render_group_fwd.o &
/opt/rocm/bin/hipcc $F -DTSG_PART=2 $BF -c $R/triangle-splatting_amd/csrc/render_group.hip -o $T/render_group_bwd.o &
wait
OBJS=""
for o in $B/*.o; do
  case $(basename $o) in
    render_group_fwd.o|render_group_bwd.o) OBJS="$OBJS $T/$(basename $o)";;
    *) OBJS="$OBJS $o";;
  esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_$TAG.so $OBJS
echo $R/tools/bin/libts2d_$TAG.so
