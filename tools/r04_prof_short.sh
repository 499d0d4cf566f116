# The measurements that depend on the blend kernels, after the quadrant-mask change: HBM traffic (separate --pmc passes), the driver's bench command,
# its rocprofv3 kernel stats.  Outputs under gpurun_out/ (tag r04q).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04q}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/pmc_gather
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_gather -- $R/tools/bin/gather_calib > $R/gpurun_out/pmc_gather.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- \
      python $R/bench.py --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_$C.log 2>&1
done
python $R/tools/hbm_traffic_summary.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE 64000000 $R/gpurun_out/pmc_gather > $R/gpurun_out/hbm_traffic.json
cp $R/gpurun_out/hbm_traffic.json $R/profiles/ 2>/dev/null
cd $R
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp
rm -rf $R/gpurun_out/prof_final
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/prof_final/*/*kernel_stats.csv | head -1) "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline" > $R/gpurun_out/${TAG}_kernel_stats.csv
find $R/gpurun_out -name "*kernel_trace.csv" -delete
find $R/gpurun_out -name "*counter_collection.csv" -size +2M -delete
tail -c 400 $R/gpurun_out/${TAG}_bench.json; head -8 $R/gpurun_out/${TAG}_kernel_stats.csv
