#!/bin/bash
# Round 4: (1) what the backward's atomic row flush costs (TSG_PROBE=6: no flush, =7: plain stores), product vs variants alternating on one
# box; (2) packed-fp32 rows of the issue-cost microbenchmark.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_probe; mkdir -p $O; cd $R
for rep in 1 2; do
for L in "" tools/bin/libts2d_noflush.so tools/bin/libts2d_storeflush.so "$@"; do
  TS2D_LIBRARY_PATH=${L:+$R/$L} timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('${L:-product}'.split('/')[-1], j['ms_per_step'], 'fwd', k['render_fwd'], 'bwd', k['render_bwd'], 'bwd_timed', j['roofline']['avg_launch_ms'])" | tee -a $O/flush.txt
done; done
tools/bin/valu_bench3 2>&1 | grep -E "pk_|v_fma_f32 |v_mul_f32|v_add_f32" | tee $O/valu_pk.txt
