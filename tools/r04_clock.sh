#!/bin/bash
# Are the kernels slower under sustained load than between the event pairs of the warm-up steps?  Per-step sequences + all kernels bracketed in the timed region.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_clock; mkdir -p $O; cd $R
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --dump-steps 2>/dev/null | tail -1 > $O/dominant.json
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --dump-steps --timed-kernel-events all 2>/dev/null | tail -1 > $O/all.json
timeout 300 python bench.py --steps 20 --warmup 5 --settle-steps 0 --no-cpu-baseline --dump-steps 2>/dev/null | tail -1 > $O/none.json
(rocm-smi --showclocks --showpower 2>/dev/null | head -40) > $O/smi_idle.txt
python - <<PY
import json
for n in ("dominant","all","none"):
    j=json.load(open("$O/%s.json"%n)); c=j["config"]
    print(n, j["ms_per_step"], "dev", c["device_step_ms"], "busy_warm", c.get("gpu_busy_ms_per_step"), "busy_timed", c.get("gpu_busy_ms_per_step_timed_region"))
    print("  seq", c["device_step_ms_all"])
    if "kernels_avg_ms_timed_region" in j: print("  timed", j["kernels_avg_ms_timed_region"]); print("  warm ", j["kernels_avg_ms"])
PY
