# Round evidence on one MI355X (gpurun): bench lines (2D headline with cpu_baseline + reference_gpu, 3D variant), rocprofv3 kernel
# stats of the same bench command, SQ counters of the blend kernels, HBM traffic (separate --pmc passes), lane-group statistics
# (stats build shipped as tools/bin/libts2d_stats.so).  Outputs under gpurun_out/; copy the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
bash $R/tools/collect_blend_pmc.sh > /dev/null 2>&1
bash $R/tools/collect_hbm_traffic.sh > /dev/null 2>&1
if [ -f $R/tools/bin/libts2d_stats.so ]; then
  L=$R/triangle-splatting_amd/diff_triangle_rasterization_2D/libts2d.so
  cp $L /tmp/keep.so; cp $R/tools/bin/libts2d_stats.so $L
  timeout 200 python $R/tests/triage/blend_probe.py > $R/gpurun_out/blend_stats.log 2>&1
  cp /tmp/keep.so $L
fi
# bench.py quotes the committed counter summaries next to its own live timings: refresh them first (copy the same files into profiles/ afterwards)
cp $R/gpurun_out/blend_pmc.json $R/gpurun_out/hbm_traffic.json $R/profiles/ 2>/dev/null
[ -f $R/gpurun_out/blend_stats.json ] && cp $R/gpurun_out/blend_stats.json $R/profiles/
timeout 300 python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
timeout 300 python $R/bench.py --rasterizer 3D > $R/gpurun_out/${TAG}_bench3d.json 2>> $R/gpurun_out/${TAG}_bench.err
rm -rf $R/gpurun_out/prof_final
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/prof_final/*/*kernel_stats.csv | head -1) "python bench.py --steps 20 --warmup 3 --no-cpu-baseline" > $R/gpurun_out/${TAG}_kernel_stats.csv
find $R/gpurun_out -name "*kernel_trace.csv" -delete
# the other BASELINE.json configurations as synthetic scenes of the same generator
: > $R/gpurun_out/${TAG}_configs.jsonl
for C in "--triangles 300000 --width 800 --height 800 --sh-degree 3" "--triangles 2000000 --width 1920 --height 1080 --sh-degree 3" \
         "--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D" "--triangles 5000000 --width 1920 --height 1080 --sh-degree 0 --rasterizer 3D" \
         "--triangles 5000000 --width 1920 --height 1080 --sh-degree 0" "--sync-free"; do
  timeout 200 python $R/bench.py $C --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_configs.jsonl
done
tail -c 400 $R/gpurun_out/${TAG}_bench.json
