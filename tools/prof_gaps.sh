cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_g
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_g -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/prof_g.log 2>&1
python $R/tools/gap_summary.py $(ls $R/gpurun_out/prof_g/*/*kernel_trace.csv | head -1) 4
find $R/gpurun_out -name "*kernel_trace.csv" -delete
