R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $R/gpurun_out/final_bench.json 2> $R/gpurun_out/final_bench.err
rm -rf $R/gpurun_out/prof_ref
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ref -- python $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/prof_ref.log 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_ref -name "*kernel_stats.csv" | head
