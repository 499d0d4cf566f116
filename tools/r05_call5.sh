#!/bin/bash
# Round 5, fifth lease: dense batches of 32 / 48 entries (one pass per batch, no second gather) against 64, forward and backward separately.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_e
mkdir -p $O
cd $R
TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_cap_fb32.so timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -k "not lab_library and not forced_ticket" > $O/pytest_fb32.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fb32.log
grep -v amdgpu.ids $O/pytest_fb32.log | grep -E "^FAILED|^ERROR|passed|failed|rc=" | head
one() { L=$1; T=$2; shift 2
  TS2D_LIBRARY_PATH=$L timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('$T', j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items() if a in ('render_fwd','render_bwd')))"
}
for rep in 1 2; do
  one "" "1M product(64)"
  for v in f32 b32 fb32 b48; do one $R/tools/bin/libts2d_cap_$v.so "1M $v"; done
done | tee $O/cap_1m.txt
for v in "" fb32; do L=""; [ -n "$v" ] && L=$R/tools/bin/libts2d_cap_$v.so; one "$L" "300k ${v:-product}" --triangles 300000 --width 800 --height 800; one "$L" "5M ${v:-product}" --triangles 5000000 --sh-degree 0; done | tee $O/cap_other.txt
