#!/bin/bash
# Round 5, fourteenth lease: is the emission bound by the number of workgroups it launches?  The same kernel with 2 / 4 consecutive blocks per workgroup.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_m
mkdir -p $O
cd $R
for i in 1 2; do
  for L in "" reps2 reps4; do
    LP=""; [ -n "$L" ] && LP=$R/tools/bin/libts2d_$L.so
    TS2D_LIBRARY_PATH=$LP timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('${L:-product}', j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items()))" | tee -a $O/reps_ab.txt
  done
done
