# The bench half of tools/final_prof.sh on a quiet box (run after the counter passes have been copied into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 python $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
timeout 300 python $R/bench.py --rasterizer 3D > $R/gpurun_out/${TAG}_bench3d.json 2>> $R/gpurun_out/${TAG}_bench.err
: > $R/gpurun_out/${TAG}_configs.jsonl
for C in "--triangles 300000 --width 800 --height 800 --sh-degree 3" "--triangles 2000000 --width 1920 --height 1080 --sh-degree 3" \
         "--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D" "--triangles 5000000 --width 1920 --height 1080 --sh-degree 0 --rasterizer 3D" \
         "--triangles 5000000 --width 1920 --height 1080 --sh-degree 0" "--sync-free"; do
  timeout 200 python $R/bench.py $C --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_configs.jsonl
done
rm -rf $R/gpurun_out/prof_final
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/prof_final/*/*kernel_stats.csv | head -1) "python bench.py --steps 20 --warmup 3 --no-cpu-baseline" > $R/gpurun_out/${TAG}_kernel_stats.csv
find $R/gpurun_out -name "*kernel_trace.csv" -delete
tail -c 300 $R/gpurun_out/${TAG}_bench.json
