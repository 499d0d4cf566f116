# per-dispatch durations of the binning kernels of one bench step (rocprofv3 --kernel-trace), in launch order
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_bin
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_bin -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
python - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
f = sorted(glob.glob(R + "/gpurun_out/trace_bin/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step only: from the last preprocess_fwd on
idx = max(i for i, r in enumerate(rows) if "preprocess_fwd" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
out = []
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0][:60]
    out.append(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))}  {n}")
open(R + "/gpurun_out/trace_binning.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find $R/gpurun_out/trace_bin -name "*.csv" -delete
