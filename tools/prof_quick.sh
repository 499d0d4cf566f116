cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_x
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_x.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/prof_x/*/*kernel_stats.csv | head -1) "bench" | head -24
find $R/gpurun_out -name "*kernel_trace.csv" -delete
