#!/bin/bash
# Round 5, ninth lease: chunk length of the instance sort picked from the capacity (2048 below 2.5 M instances): A/B through the lab environment word.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_i
mkdir -p $O
cd $R
run() { # tag, env value, bench args
  local tag=$1 v=$2; shift 2
  TS2D_LAB_INSTANCE_SMALL_BELOW=$v timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('$tag below=$v', j['ms_per_step'], 'tile_sort=%.4f emit=%.4f ranges=%.4f' % (k['tile_sort'], k['emit_keys'], k['tile_ranges']))" | tee -a $O/ab.txt
}
for i in 1 2; do
  for v in 0 2500000; do
    run 10k $v --triangles 10000 --width 256 --height 256 --sh-degree 0 --hip-graph --steps 200 --warmup 20
    run 93k3d $v --triangles 93000 --width 1600 --height 1600 --rasterizer 3D --hip-graph --steps 100 --warmup 10
    run 300k $v --triangles 300000 --width 800 --height 800 --hip-graph --steps 100 --warmup 10
  done
done
for v in 0 5000000; do run 1M $v --steps 20 --warmup 5; done
timeout 900 python -m pytest tests/test_reference_gpu.py tests/test_parity_gpu.py tests/test_speculative_forward_gpu.py tests/test_async_forward_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|^E  |passed|failed|rc=" | head -20
