# Round-5 evidence on one MI355X (gpurun): rocprofv3 kernel stats of the DRIVER'S bench command, HBM traffic (separate --pmc passes), SQ
# counters of the blend kernels, the bench line (2D headline with cpu_baseline + reference_gpu), the 3D variant, the other BASELINE
# configurations, the headline with the fourth depth pass forced / with the fused Adam step / sync-free.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
cd $R
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/${TAG}_pytest.log
grep -v amdgpu.ids $R/gpurun_out/${TAG}_pytest.log | grep -E "^FAILED|^ERROR|passed|failed|rc=" | head -30
cd /tmp && export TMPDIR=/tmp
bash $R/tools/collect_blend_pmc.sh > /dev/null 2>&1
rm -rf $R/gpurun_out/pmc_gather
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_gather -- $R/tools/bin/gather_calib > $R/gpurun_out/pmc_gather.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- \
      python $R/bench.py --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_$C.log 2>&1
done
python $R/tools/hbm_traffic_summary.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE 64000000 $R/gpurun_out/pmc_gather > $R/gpurun_out/hbm_traffic.json
[ -f $R/tools/bin/libts2d_stats.so ] && TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_stats.so timeout 200 python $R/tests/triage/blend_probe.py > $R/gpurun_out/blend_stats.log 2>&1
cp $R/gpurun_out/blend_pmc.json $R/gpurun_out/hbm_traffic.json $R/profiles/ 2>/dev/null
[ -f $R/gpurun_out/blend_stats.json ] && cp $R/gpurun_out/blend_stats.json $R/profiles/
cd $R
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
for i in 1 2 3; do timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_bench_series.jsonl; done
timeout 300 python $R/bench.py --rasterizer 3D > $R/gpurun_out/${TAG}_bench3d.json 2>> $R/gpurun_out/${TAG}_bench.err
cd /tmp
rm -rf $R/gpurun_out/prof_final
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/prof_final/*/*kernel_stats.csv | head -1) "python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline" > $R/gpurun_out/${TAG}_kernel_stats.csv
python $R/tools/gap_summary.py $(ls $R/gpurun_out/prof_final/*/*kernel_trace.csv | head -1) 4 > $R/gpurun_out/${TAG}_gaps.txt 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -delete
cd $R
: > $R/gpurun_out/${TAG}_configs.jsonl
for C in "--triangles 10000 --width 256 --height 256 --sh-degree 0" "--triangles 300000 --width 800 --height 800 --sh-degree 3" "--triangles 2000000 --width 1920 --height 1080 --sh-degree 3" \
         "--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D" "--triangles 5000000 --width 1920 --height 1080 --sh-degree 0 --rasterizer 3D" \
         "--triangles 5000000 --width 1920 --height 1080 --sh-degree 0" "--sync-free" "--with-optimizer" "--settle-steps 0" "--hip-graph" \
         "--triangles 10000 --width 256 --height 256 --sh-degree 0 --hip-graph" "--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --hip-graph" \
         "--triangles 300000 --width 800 --height 800 --sh-degree 3 --hip-graph" "--gamma 50" "--gamma 7" "--rasterizer 3D --gamma 50" \
         "--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --gamma 50 --hip-graph" "--scene-mode centered"; do
  timeout 200 python $R/bench.py $C --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_configs.jsonl
done
TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_lab.so timeout 200 python $R/bench.py --force-depth-pass4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $R/gpurun_out/${TAG}_configs.jsonl
bash $R/tools/prof_gaps.sh --triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D > $R/gpurun_out/${TAG}_gaps_93k3d.txt 2>&1
python - <<PY
import json
for f in ("$R/gpurun_out/${TAG}_bench_series.jsonl", "$R/gpurun_out/${TAG}_configs.jsonl"):
    for l in open(f):
        try: j=json.loads(l)
        except Exception: print("BAD", l[:200]); continue
        c=j["config"]; k=j.get("kernels_avg_ms",{})
        print(c["rasterizer"], c["triangles"], c["width"], c["workload"].split(":")[0][-24:], c["forward"][:10], "ms", j["ms_per_step"], "host", c["host_step_ms"]["median"], "idle", c.get("gpu_idle_ms_per_step"), " ".join(f"{a}={b:.4f}" for a,b in k.items()))
PY
tail -c 600 $R/gpurun_out/${TAG}_bench.json
