#!/bin/bash
# Round 5, seventeenth lease: ballot-xor-or match in the multi-launch scatter + dwordx4 loads in the histogram: tests, then A/B against the previous commit's library.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_binning_gpu.py tests/test_reference_gpu.py tests/test_knn_gpu.py tests/test_parity_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -2
run() {
  local tag=$1 L=$2; shift 2
  LP=""; [ -n "$L" ] && LP=$R/tools/bin/libts2d_$L.so
  TS2D_LIBRARY_PATH=$LP timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('$tag ${L:-product}', j['ms_per_step'], 'census=%.4f depth=%.4f scan=%.4f emit=%.4f tile_sort=%.4f ranges=%.4f' % (k['depth_census'], k['depth_sort'], k['scan'], k['emit_keys'], k['tile_sort'], k['tile_ranges']))" | tee -a $O/match_ab.txt
}
for i in 1 2; do
  for L in "" prev; do
    run 1M "$L" --steps 20 --warmup 5
    run 93k3d "$L" --triangles 93000 --width 1600 --height 1600 --rasterizer 3D --hip-graph --steps 100 --warmup 10
    run 300k "$L" --triangles 300000 --width 800 --height 800 --hip-graph --steps 100 --warmup 10
    run 5M "$L" --triangles 5000000 --sh-degree 0 --steps 10 --warmup 3
  done
done
