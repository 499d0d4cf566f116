#!/bin/bash
# Round 5, second lease: affine quadrant masks in the emission (2D + 3D), dense batches in the 3D blend kernels, the fused statistics network of the
# forward, HIP-graph replay of the step: full suite, product vs the previous commit's library alternating on one box, the other configurations.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|rc=" | head -40
echo "== 2D headline: product / previous alternating"; bash tools/ab_bench.sh $R/tools/bin/libts2d_prev.so --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tee $O/ab_2d.txt
echo "== 3D headline: product / previous alternating"; bash tools/ab_bench.sh $R/tools/bin/libts2d_prev.so --steps 20 --warmup 5 --rasterizer 3D 2>&1 | grep -v amdgpu.ids | tee $O/ab_3d.txt
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
S3="--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D"
S0="--triangles 10000 --width 256 --height 256 --sh-degree 0"
for extra in "" "--sync-free" "--hip-graph"; do
  $B $S3 $extra 2>$O/err.txt | tail -1 >> $O/configs.jsonl || tail -3 $O/err.txt
  $B $S0 $extra 2>$O/err.txt | tail -1 >> $O/configs.jsonl || tail -3 $O/err.txt
done
$B --triangles 300000 --width 800 --height 800 --hip-graph 2>/dev/null | tail -1 >> $O/configs.jsonl
$B --hip-graph 2>/dev/null | tail -1 >> $O/configs.jsonl
$B --triangles 5000000 --sh-degree 0 2>/dev/null | tail -1 >> $O/configs.jsonl
$B --triangles 5000000 --sh-degree 0 --rasterizer 3D 2>/dev/null | tail -1 >> $O/configs.jsonl
TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_prev.so $B --triangles 5000000 --sh-degree 0 2>/dev/null | tail -1 >> $O/configs_prev.jsonl
TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_prev.so $B --triangles 5000000 --sh-degree 0 --rasterizer 3D 2>/dev/null | tail -1 >> $O/configs_prev.jsonl
TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_prev.so $B $S3 2>/dev/null | tail -1 >> $O/configs_prev.jsonl
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.jsonl")):
    for l in open(f):
        try: j=json.loads(l)
        except Exception: print(f, "BAD", l[:200]); continue
        c=j["config"]; k=j.get("kernels_avg_ms",{})
        print(f.split("/")[-1], c["rasterizer"], c["triangles"], c["width"], c["forward"][:12], "ms", j["ms_per_step"], "host", c["host_step_ms"]["median"], "idle", c.get("gpu_idle_ms_per_step"),
              " ".join(f"{a}={b:.4f}" for a,b in k.items()))
PY
