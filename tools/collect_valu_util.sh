#!/bin/bash
# VALU utilisation of the blend kernels from SQ counters (own --pmc pass, --kernel-trace only):
#   SQ_ACTIVE_INST_VALU (quad-cycles in which a SIMD's VALU is executing), SQ_INSTS_VALU, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE.
# Output: gpurun_out/valu_util.json (copy to profiles/).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_valu
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_valu -- \
    python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_valu.log 2>&1
python - "$R" <<'PY'
import csv, glob, json, re, sys, collections
R = sys.argv[1]
f = sorted(glob.glob(R + "/gpurun_out/pmc_valu/*/*counter_collection.csv"))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"(render_fwd_kernel|render_bwd_kernel|render3d_fwd_kernel|render3d_bwd_kernel)", r["Kernel_Name"])
    if m:
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    simds = 256 * 4
    gui = mean.get("GRBM_GUI_ACTIVE", 0.0)
    out[k] = {"launches": len(next(iter(c.values()))), **{n: round(v, 1) for n, v in mean.items()},
              # VALU-active cycles per SIMD (the counter is in quad-cycles, summed over all SIMDs) over the kernel's cycles
              # (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  Values slightly above 1 are possible: the counter integrates
              # per-wave activity and consecutive instructions overlap in the pipeline.
              "valu_busy_frac": round(4.0 * mean.get("SQ_ACTIVE_INST_VALU", 0.0) / simds / (gui / 8.0), 4) if gui else None,
              "valu_insts_per_launch": round(mean.get("SQ_INSTS_VALU", 0.0)),
              "cycles_per_valu_inst": round(4.0 * mean.get("SQ_ACTIVE_INST_VALU", 0.0) / mean["SQ_INSTS_VALU"], 2) if mean.get("SQ_INSTS_VALU") else None}
json.dump(out, open(R + "/gpurun_out/valu_util.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
