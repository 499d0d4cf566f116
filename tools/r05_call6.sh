#!/bin/bash
# Round 5, sixth lease: machine-scheduler strategies for the blend kernels (render_group.hip recompiled with -mllvm -amdgpu-sched-strategy=...).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_f
mkdir -p $O
cd $R
one() { L=$1; T=$2; shift 2
  TS2D_LIBRARY_PATH=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('$T', j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items() if a in ('render_fwd','render_bwd')))"
}
for rep in 1 2; do
  one "" "1M product"
  for v in max-ilp iterative-ilp iterative-minreg max-memory-clause; do one $R/tools/bin/libts2d_sched_$v.so "1M $v"; done
done | tee $O/sched_1m.txt
