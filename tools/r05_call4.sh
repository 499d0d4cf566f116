#!/bin/bash
# Round 5, fourth lease: emission variants (register budget x record prefetch) on three scenes, the 3D backward at 6 waves per SIMD, GraphedStep tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_async_forward_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|^E  |passed|failed|rc=" | head -30
one() { # lib, tag, bench args...
  L=$1; T=$2; shift 2
  TS2D_LIBRARY_PATH=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('$T', j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items() if a in ('emit_keys','render_fwd','render_bwd','scan')))"
}
V="w0p1 w0p0 w6p1 w6p0 w7p0"
for rep in 1 2; do for v in $V; do one $R/tools/bin/libts2d_emit_$v.so "1M-2D $v"; done; done | tee $O/emit_1m.txt
for v in $V; do one $R/tools/bin/libts2d_emit_$v.so "5M-2D $v" --triangles 5000000 --sh-degree 0; done | tee $O/emit_5m.txt
for v in $V; do one $R/tools/bin/libts2d_emit_$v.so "1M-3D $v" --rasterizer 3D; done | tee $O/emit_3d.txt
for v in w0p0 w6p1; do one $R/tools/bin/libts2d_emit_$v.so "5M-3D $v" --triangles 5000000 --sh-degree 0 --rasterizer 3D; done | tee -a $O/emit_3d.txt
for rep in 1 2; do
  one "" "1M-3D product(bwd5)" --rasterizer 3D; one $R/tools/bin/libts2d_r3bwd6.so "1M-3D bwd6" --rasterizer 3D
done | tee $O/r3bwd.txt
one "" "93k-3D product(bwd5)" --triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --hip-graph | tee -a $O/r3bwd.txt
one $R/tools/bin/libts2d_r3bwd6.so "93k-3D bwd6" --triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --hip-graph | tee -a $O/r3bwd.txt
