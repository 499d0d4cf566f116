#!/bin/bash
# Statistics build of the blend kernels (-DTS2D_STATS) -> tools/bin/libts2d_stats.so: the lab library's objects with render_q8 / render_group
# recompiled with counters.  Used by tests/triage/q8_probe.py and blend_probe.py (TS2D_LIBRARY_PATH=tools/bin/libts2d_stats.so).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
python $R/triangle-splatting_amd/build.py --lab > /dev/null
B=$R/triangle-splatting_amd/build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG -DTS2D_STATS -mllvm -amdgpu-atomic-optimizer-strategy=None -fno-slp-vectorize"
mkdir -p $R/tools/bin /tmp/ts2d_stats
C=$R/triangle-splatting_amd/csrc
/opt/rocm/bin/hipcc $F -I$C -c $R/tools/lab/render_q8.hip -o /tmp/ts2d_stats/render_q8.o &
/opt/rocm/bin/hipcc $F -DTSG_PART=1 -c $C/render_group.hip -o /tmp/ts2d_stats/render_group_fwd.o &   # (two translation units since round 6)
/opt/rocm/bin/hipcc $F -DTSG_PART=2 -c $C/render_group.hip -o /tmp/ts2d_stats/render_group_bwd.o &
wait
OBJS=""
for o in $B/lab/*.o $B/*.o; do
  n=$(basename $o)
  case " $SEEN " in *" $n "*) continue;; esac   # lab/api.o replaces api.o
  SEEN="$SEEN $n"
  if [ -f /tmp/ts2d_stats/$n ]; then OBJS="$OBJS /tmp/ts2d_stats/$n"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/libts2d_stats.so $OBJS
echo $R/tools/bin/libts2d_stats.so
