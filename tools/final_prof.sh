# Round-end evidence: bench line (with cpu_baseline + reference_gpu) and rocprofv3 kernel stats for the 2D headline and the
# 3D variant; PMC HBM traffic with tools/collect_hbm_traffic.sh (separate passes).  Outputs under gpurun_out/; the summaries
# are condensed into profiles/ by tools/rocprof_summary.py.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 python $R/bench.py > $R/gpurun_out/final_bench.json 2> $R/gpurun_out/final_bench.err
timeout 300 python $R/bench.py --rasterizer 3D > $R/gpurun_out/final_bench3d.json 2>> $R/gpurun_out/final_bench.err
rm -rf $R/gpurun_out/prof_final $R/gpurun_out/prof_final3d
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final3d -- python $R/bench.py --rasterizer 3D --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final3d.log 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -delete
tail -c 300 $R/gpurun_out/final_bench.json
