#!/bin/bash
# Round 5, sixteenth lease: 1024-pair chunks adopted for sorts of up to 2.5 M pairs: suite + the configurations.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_o
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|^E  |passed|failed|rc=" | head -20
run() {
  local tag=$1; shift
  timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | tee -a $O/configs.jsonl | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('$tag', j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items()))"
}
run 10k --triangles 10000 --width 256 --height 256 --sh-degree 0 --hip-graph --steps 200 --warmup 20
run 93k3d --triangles 93000 --width 1600 --height 1600 --rasterizer 3D --hip-graph --steps 100 --warmup 10
run 300k --triangles 300000 --width 800 --height 800 --hip-graph --steps 100 --warmup 10
run 1M --steps 20 --warmup 5
run 1M --steps 20 --warmup 5
run 1M-3D --rasterizer 3D --steps 20 --warmup 5
run 2M --triangles 2000000 --steps 20 --warmup 5
run 5M-3D --triangles 5000000 --sh-degree 0 --rasterizer 3D --steps 10 --warmup 3
run 5M --triangles 5000000 --sh-degree 0 --steps 10 --warmup 3
