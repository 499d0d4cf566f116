#!/bin/bash
# Round 5, fifteenth lease: 1024-pair chunks for the small sorts (variant library) against the product's 2048, alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_n
mkdir -p $O
cd $R
run() {
  local tag=$1 L=$2; shift 2
  LP=""; [ -n "$L" ] && LP=$R/tools/bin/libts2d_$L.so
  TS2D_LIBRARY_PATH=$LP timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['kernels_avg_ms']
print('$tag ${L:-product}', j['ms_per_step'], 'census=%.4f depth=%.4f scan=%.4f emit=%.4f tile_sort=%.4f' % (k['depth_census'], k['depth_sort'], k['scan'], k['emit_keys'], k['tile_sort']))" | tee -a $O/ch1024_ab.txt
}
for i in 1 2; do
  for L in ch1024 ch512; do
    run 93k3d "$L" --triangles 93000 --width 1600 --height 1600 --rasterizer 3D --hip-graph --steps 100 --warmup 10
    run 300k "$L" --triangles 300000 --width 800 --height 800 --hip-graph --steps 100 --warmup 10
    run 10k "$L" --triangles 10000 --width 256 --height 256 --sh-degree 0 --hip-graph --steps 200 --warmup 20
    run 1M "$L" --steps 20 --warmup 5
  done
done
