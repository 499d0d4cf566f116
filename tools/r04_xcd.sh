#!/bin/bash
# Round 4: blockIdx -> tile mapping over the 8 XCDs.  rows (product) vs contiguous bands (rounds 1-3, -DTS2D_XCD_BANDS) on the uniform headline scene
# and on the same triangles concentrated about the optical axis (--scene-mode centered), alternating on one box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_xcd; mkdir -p $O; cd $R
for rep in 1 2; do
for MODE in frustum centered; do
for L in "" tools/bin/libts2d_bands.so; do
  TS2D_LIBRARY_PATH=${L:+$R/$L} timeout 200 python bench.py --no-cpu-baseline --scene-mode $MODE 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('$MODE', '${L:-rows(product)}'.split('/')[-1], 'N', j['config']['num_rendered'], 'ms/step', j['ms_per_step'], 'fwd', k['render_fwd'], 'bwd', k['render_bwd'])" | tee -a $O/xcd.txt
done; done; done
