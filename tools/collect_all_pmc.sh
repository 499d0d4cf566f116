#!/bin/bash
# SQ counters of EVERY kernel of a step on the headline scene (same passes as collect_blend_pmc.sh, output gpurun_out/all_pmc.json)
# SQ counters of the blend kernels on the headline scene (two --pmc passes of 8 SQ counters + GRBM_GUI_ACTIVE, --kernel-trace only).
# Output: gpurun_out/all_pmc.json (copy to profiles/).  Units: SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed
# over waves (MI355X_MICROARCH.md "Per-instruction cycle constants"); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_all$i
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_all$i -- python $R/tests/triage/blend_probe.py > $R/gpurun_out/pmc_all$i.log 2>&1
done
python - "$R" <<'PY'
import csv, glob, json, re, sys, collections
R = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    fs = sorted(glob.glob(R + f"/gpurun_out/pmc_all{i}/*/*counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        m = re.search(r"(\w*_kernel)<(\w+), (\w+)", r["Kernel_Name"]) or re.search(r"(\w*_kernel)", r["Kernel_Name"])
        if m:
            name = m.group(1) + ("<rich>" if len(m.groups()) > 1 and m.group(2) == "true" else "<plain>" if len(m.groups()) > 1 else "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0  # cycles of the launch
    simd_cycles = 1024 * gui
    d = {"launches": len(next(iter(c.values()))), "kernel_cycles": round(gui), **{n: round(v) for n, v in m.items()}}
    if "SQ_INSTS_VALU" in m and gui:
        d["valu_issue_frac_at_2cyc"] = round(2 * m["SQ_INSTS_VALU"] / simd_cycles, 4)
        d["lane_occupancy"] = round(m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_ACTIVE_INST_VALU"]), 4) if m.get("SQ_ACTIVE_INST_VALU") and m.get("SQ_THREAD_CYCLES_VALU") else None
        d["lds_busy_frac"] = round(m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * gui), 4)
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        d["wave_cycle_split"] = {n: round(m.get(n, 0) / w, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")}
        d["avg_waves_per_simd"] = round(4 * w / simd_cycles, 3) if gui else None
    out[k] = d
json.dump(out, open(R + "/gpurun_out/all_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
