set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 python $R/bench.py > $R/gpurun_out/final_bench.json 2> $R/gpurun_out/final_bench.err
tail -c 600 $R/gpurun_out/final_bench.json
rm -rf $R/gpurun_out/prof_final $R/gpurun_out/prof_final3d
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final3d -- python $R/bench.py --rasterizer 3D --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final3d.log 2>&1
find $R/gpurun_out/prof_final $R/gpurun_out/prof_final3d -name "*kernel_stats.csv" | head
timeout 500 bash $R/tools/collect_hbm_traffic.sh > $R/gpurun_out/hbm_collect.log 2>&1
tail -5 $R/gpurun_out/hbm_collect.log
# keep only the small summaries
find $R/gpurun_out -name "*kernel_trace.csv" -delete
find $R/gpurun_out -name "*counter_collection.csv" -size +20M -delete
du -sh $R/gpurun_out
