#!/bin/bash
# Round 4: the driver's bench command, N fresh processes in a row on one lease (+ variants), and the idle-time table of one traced run.
# usage: tools/r04_runs.sh <tag> [tests]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}
O=$R/gpurun_out/r04_$TAG
mkdir -p $O
cd $R
if [ "$2" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  tail -5 $O/pytest.log
fi
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/default.jsonl
done
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --settle-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/nosettle.jsonl
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --sync-free --no-cpu-baseline 2>/dev/null | tail -1 >> $O/syncfree.jsonl
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.jsonl")):
    for l in open(f):
        try: j=json.loads(l)
        except Exception: print(f, "BAD", l[:200]); continue
        c=j["config"]; k=j.get("kernels_avg_ms",{})
        print(f.split("/")[-1], j["ms_per_step"], j["value"], "host",c["host_step_ms"], "dev",c.get("device_step_ms"), "idle",c.get("gpu_idle_ms_per_step"), "busy", c.get("gpu_busy_ms_per_step"), "bwd_in_run", j["roofline"]["avg_launch_ms"] if j.get("roofline") else None, "fwd", k.get("render_fwd"), "bwd", k.get("render_bwd"))
PY
bash tools/prof_gaps.sh > $O/gaps.txt 2>&1; tail -30 $O/gaps.txt
