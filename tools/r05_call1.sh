#!/bin/bash
# Round 5, first lease: the full GPU suite with the new pins (gamma > 1 at size, the reference's uncontracted 2D build, ShardedAdam through the
# rasterizer), the driver's command, and bench lines for gamma > 1 / the other configurations (profiles/r05_gamma.jsonl, r05_configs.jsonl).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | tail -40
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5"
$B 2>/dev/null | tail -1 > $O/default.jsonl
for g in 7 50; do $B --no-cpu-baseline --gamma $g 2>/dev/null | tail -1 >> $O/gamma.jsonl; done
$B --no-cpu-baseline --rasterizer 3D 2>/dev/null | tail -1 >> $O/gamma.jsonl
$B --no-cpu-baseline --rasterizer 3D --gamma 50 2>/dev/null | tail -1 >> $O/gamma.jsonl
S="--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D"
$B --no-cpu-baseline $S 2>/dev/null | tail -1 >> $O/gamma.jsonl
$B --no-cpu-baseline $S --gamma 50 2>/dev/null | tail -1 >> $O/gamma.jsonl
$B --no-cpu-baseline --triangles 10000 --width 256 --height 256 --sh-degree 0 2>/dev/null | tail -1 >> $O/configs.jsonl
$B --no-cpu-baseline --triangles 300000 --width 800 --height 800 2>/dev/null | tail -1 >> $O/configs.jsonl
$B --no-cpu-baseline --triangles 5000000 --sh-degree 0 2>/dev/null | tail -1 >> $O/configs.jsonl
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.jsonl")):
    for l in open(f):
        try: j=json.loads(l)
        except Exception: print(f, "BAD", l[:200]); continue
        c=j["config"]; k=j.get("kernels_avg_ms",{})
        print(f.split("/")[-1], c["rasterizer"], c["triangles"], c["width"], "|", c["workload"].split(":")[0][-22:], "ms", j["ms_per_step"], "host", c["host_step_ms"]["median"], "idle", c.get("gpu_idle_ms_per_step"),
              " ".join(f"{a}={b:.4f}" for a,b in k.items()))
PY
