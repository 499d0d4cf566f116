#!/bin/bash
# Round 5, eighteenth lease: forward with the next batch's list windows read under the current batch's record gather (A/B).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_q
mkdir -p $O
cd $R
for i in 1 2; do
  for L in "" ahead ahead6; do
    LP=""; [ -n "$L" ] && LP=$R/tools/bin/libts2d_$L.so
    TS2D_LIBRARY_PATH=$LP timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('${L:-product}', j['ms_per_step'], 'fwd=%.4f bwd=%.4f' % (k['render_fwd'], k['render_bwd']))" | tee -a $O/ahead_ab.txt
  done
done
TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_ahead.so timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "not lab" 2>&1 | grep -v amdgpu.ids | tail -2
