#!/bin/bash
# Round 5, twelfth lease: non-temporal stores / loads for the staged rows of the per-triangle kernels (A/B).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_l
mkdir -p $O
cd $R
for i in 1 2; do
  for L in "" nt1 nt2 nt3; do
    LP=""; [ -n "$L" ] && LP=$R/tools/bin/libts2d_$L.so
    TS2D_LIBRARY_PATH=$LP timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('${L:-product}', j['ms_per_step'], 'pre_fwd=%.4f pre_bwd=%.4f emit=%.4f fwd=%.4f bwd=%.4f' % (k['preprocess_fwd'], k['preprocess_bwd'], k['emit_keys'], k['render_fwd'], k['render_bwd']))" | tee -a $O/nt_ab.txt
  done
done
