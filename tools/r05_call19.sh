#!/bin/bash
# Round 5, nineteenth lease: rectangle + tile count in one 16-byte element, rectangles written in depth order by the scan's gather and read in order by the emission.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_binning_gpu.py tests/test_reference_gpu.py tests/test_parity_gpu.py tests/test_parity3d_gpu.py tests/test_speculative_forward_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -2
run() {
  local tag=$1 L=$2; shift 2
  LP=""; [ -n "$L" ] && LP=$R/tools/bin/libts2d_$L.so
  TS2D_LIBRARY_PATH=$LP timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('$tag ${L:-product}', j['ms_per_step'], 'pre_fwd=%.4f census=%.4f scan=%.4f emit=%.4f tile_sort=%.4f' % (k['preprocess_fwd'], k['depth_census'], k['scan'], k['emit_keys'], k['tile_sort']))" | tee -a $O/rect_ab.txt
}
for i in 1 2; do
  for L in "" prev; do
    run 1M "$L" --steps 20 --warmup 5
    run 2M "$L" --triangles 2000000 --steps 15 --warmup 4
    run 5M "$L" --triangles 5000000 --sh-degree 0 --steps 10 --warmup 3
    run 5M3D "$L" --triangles 5000000 --sh-degree 0 --rasterizer 3D --steps 10 --warmup 3
    run 300k "$L" --triangles 300000 --width 800 --height 800 --hip-graph --steps 100 --warmup 10
  done
done
