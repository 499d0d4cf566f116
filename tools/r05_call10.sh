#!/bin/bash
# Round 5, tenth lease: one-launch depth order for small scenes (depth_order_small_kernel): suite + small-scene steps.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|^E  |passed|failed|rc=" | head -20
run() {
  local tag=$1; shift
  timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | tee -a $O/small.jsonl | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('$tag', j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items()))"
}
for i in 1 2; do
  run 10k --triangles 10000 --width 256 --height 256 --sh-degree 0 --hip-graph --steps 200 --warmup 20
  run 16k --triangles 16384 --width 256 --height 256 --sh-degree 0 --hip-graph --steps 200 --warmup 20
  run 17k --triangles 16385 --width 256 --height 256 --sh-degree 0 --hip-graph --steps 200 --warmup 20
done
run 10k-eager --triangles 10000 --width 256 --height 256 --sh-degree 0 --steps 200 --warmup 20
