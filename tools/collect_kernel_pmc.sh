#!/bin/bash
# SQ counters of ANY kernels of the step whose name matches a regular expression (two --pmc passes, --kernel-trace only), workload: bench.py.
#   usage: tools/collect_kernel_pmc.sh '<regex>' <out.json> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
RE=$1; OUT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_k$i
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_k$i -- python $R/bench.py --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-kernel-events "$@" > $R/gpurun_out/pmc_k$i.log 2>&1
done
python - "$R" "$RE" "$OUT" <<'PY'
import csv, glob, json, re, sys, collections
R, RE, OUT = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    fs = sorted(glob.glob(R + f"/gpurun_out/pmc_k{i}/*/*counter_collection.csv"))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if re.search(RE, r["Kernel_Name"]):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")[:70]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    simd_cycles = 1024 * gui
    d = {"launches": len(next(iter(c.values()))), "kernel_cycles": round(gui), **{n: round(v) for n, v in m.items()}}
    if "SQ_WAVE_CYCLES" in m and gui:
        w = m["SQ_WAVE_CYCLES"]
        d["wave_cycle_split"] = {n: round(m.get(n, 0) / w, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA")}
        d["avg_waves_per_simd"] = round(4 * w / simd_cycles, 3)
    if "SQ_INSTS_VALU" in m and gui:
        d["valu_issue_frac_at_2cyc"] = round(2 * m["SQ_INSTS_VALU"] / simd_cycles, 4)
    out[k] = d
json.dump(out, open(OUT, "w"), indent=1)
for k, d in out.items():
    print(k, d.get("kernel_cycles"), d.get("avg_waves_per_simd"), d.get("wave_cycle_split"), {n: d[n] for n in d if n.startswith("SQ_INSTS") or n.startswith("SQ_WAIT_INST_LDS") or n.startswith("SQ_ACTIVE_INST_VMEM") or n.startswith("SQ_LDS")})
PY
