#!/bin/bash
# tools/lease.sh -- ONE parameterised script for every GPU lease (round 6 on).  Rounds 3-5 kept one script per lease (tools/r03_prof.sh, r04_*.sh,
# r05_call1.sh ... r05_call19.sh, final_*.sh: ~30 files, deleted in round 6 -- `git log -- tools/r05_call7.sh` has them; every one was a sequence of
# the steps below: tests, bench lines, A/B of two libraries, rocprofv3 trace, counter pass).  A lease is a list of steps, each a shell function:
#     gpurun --timeout 1500 -- 'bash tools/lease.sh r06_a "t tests/test_side_stream_gpu.py; bench head; ab r05 2; prof head"'
# Output goes to gpurun_out/<tag>/ (merged back by gpurun); what is worth judging is copied from there into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp

# t <pytest args>: GPU tests, quiet, last lines only
t() { timeout 1500 python -m pytest "$@" -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -40 | tee -a $O/tests.txt; }
# lab <name> [bench.py args]: bench.py on the lab library
lab() { local n=$1; shift; TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_lab.so bench $n --no-cpu-baseline "$@"; }

# summary of one bench line on stdin: ms/step + the kernels' averages
_sum() { python -c "
import json,sys
L=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
if not L: print('$1 NO LINE'); sys.exit()
j=json.loads(L[-1]); k=j.get('kernels_avg_ms',{}); c=j['config']
print('$1', j['ms_per_step'], 'dev_med=%s idle=%s host_med=%s' % (c['device_step_ms']['median'], c.get('gpu_idle_ms_per_step'), c['host_step_ms']['median']), ' '.join(f'{a}={b:.4f}' for a,b in k.items()))"; }

# bench <name> [bench.py args]: one run of bench.py, the JSON line appended to <name>.jsonl
bench() { local n=$1; shift; timeout 600 python bench.py "$@" 2>$O/$n.err | tee -a $O/$n.jsonl | _sum $n | tee -a $O/summary.txt; }

# ab <libtag> <rounds> [bench.py args]: the product library and tools/bin/libts2d_<libtag>.so alternating on this box
ab() { local L=$1 n=$2; shift 2; export TS2D_BINDING=ctypes; for i in $(seq $n); do   # both sides through ctypes: the compiled binding is linked to the product library
    bench ab_product --no-cpu-baseline "$@"
    TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_$L.so bench ab_$L --no-cpu-baseline "$@"; done; unset TS2D_BINDING; }

# abflag <rounds> <flag> [bench.py args]: the lab library with and without one bench.py switch (e.g. --no-side-stream), alternating
abflag() { local n=$1 f=$2; shift 2; for i in $(seq $n); do
    TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_lab.so bench abflag_off --no-cpu-baseline "$@"
    TS2D_LIBRARY_PATH=$R/tools/bin/libts2d_lab.so bench "abflag_${f#--}" --no-cpu-baseline $f "$@"; done; }

# abprod <rounds> <flag> [bench.py args]: the PRODUCT library with and without one bench.py switch (e.g. --no-prepare-backward), alternating
abprod() { local n=$1 f=$2; shift 2; for i in $(seq $n); do
    bench abprod_off --no-cpu-baseline "$@"
    bench "abprod_${f#--}" --no-cpu-baseline $f "$@"; done; }

# prof <name> [bench.py args]: rocprofv3 kernel trace + stats of bench.py (no counters in this pass)
prof() { local n=$1; shift; rm -rf $O/prof_$n; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o $n --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $O/prof_$n.log 2>&1)
    f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${n}_kernel_stats.csv && head -25 $f
    tr=$(find $O/prof_$n -name "*kernel_trace.csv" | head -1); [ -n "$tr" ] && python $R/tools/overlap_summary.py $tr | tee $O/${n}_overlap.txt
    find $O/prof_$n -name "*kernel_trace.csv" -size +20M -delete; }

# pmc <name> "<counters>" [bench.py args]: one counter pass (own run, kernel trace only -- gpurun refuses counters with other trace domains)
pmc() { local n=$1 c=$2; shift 2; rm -rf $O/pmc_$n; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n -o $n --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 --settle-steps 0 --no-kernel-events "$@" > $O/pmc_$n.log 2>&1)
    f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $f | tee $O/${n}_pmc.txt; rm -rf $O/pmc_$n; }

# pmcx <name> "<counters>" <command...>: one counter pass over ANY command (kernel trace only, no other trace domain)
pmcx() { local n=$1 c=$2; shift 2; rm -rf $O/pmc_$n; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$n -o $n --output-format csv -- "$@" > $O/pmc_$n.log 2>&1)
    f=$(find $O/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $f 3 | tee $O/${n}_pmc.txt; rm -rf $O/pmc_$n; }
# train <config> [args]: one line of tools/bench_train_step.py appended to train_step.jsonl
train() { timeout 900 python tools/bench_train_step.py --config "$@" 2>$O/train_$1.err | tee -a $O/train_step.jsonl | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); k=j['kernels']
    print(j['config'], 'eager', j['iteration_ms_eager'], 'eager_sync_free', j.get('iteration_ms_eager_sync_free'), 'graph', j['iteration_ms_graph'], j['graph_note'] or '', 'raster', j['raster_step_ms'], 'diff', j['iteration_minus_raster_ms'], 'glue', j['torch_glue_ms_per_iteration'])
    print('   ', ' '.join('%s=%.4f(%s)' % (n, e['avg_ms'], e.get('hbm_frac','-')) for n,e in k.items() if n not in ('depth_census','depth_sort','scan','emit_keys','tile_sort','tile_ranges')))"; }
# loss [H W]: tools/bench_loss.py (the fused L1 + SSIM against the eager-torch kernel sequence)
loss() { timeout 600 python tools/bench_loss.py "$@" 2>/dev/null | tee -a $O/loss_bench.jsonl; }

# final: everything a round's profiles/ needs from ONE box, with the committed tree -- the driver's command four times, its rocprofv3 kernel stats and
# gaps, counter passes (HBM traffic with both calibrations, SQ counters of the blend kernels), the other configurations, forward-only lines, training
# iterations, the loss kernels.  Copy what is worth judging from gpurun_out/<tag>/ into profiles/ (named r<NN>_*).
final() {
    t tests/test_side_stream_gpu.py tests/test_loss_gpu.py tests/test_model_init_gpu.py tests/test_binding_gpu.py
    for i in 1 2 3 4; do bench bench_series --steps 20 --warmup 5 $([ $i -gt 1 ] && echo --no-cpu-baseline); done
    prof driver --steps 20 --warmup 5
    bash tools/prof_gaps.sh > $O/gaps.txt 2>&1; tail -20 $O/gaps.txt
    bash tools/collect_hbm_traffic.sh > $O/hbm_traffic.log 2>&1; cp gpurun_out/hbm_traffic.json $O/ 2>/dev/null; tail -3 $O/hbm_traffic.log
    bash tools/collect_blend_pmc.sh > $O/blend_pmc.log 2>&1; cp gpurun_out/blend_pmc.json $O/ 2>/dev/null; tail -3 $O/blend_pmc.log
    c() { local n=$1; shift; bench configs --no-cpu-baseline "$@"; }
    c 10k --triangles 10000 --width 256 --height 256 --sh-degree 0 --steps 200 --warmup 20
    c 10k_graph --triangles 10000 --width 256 --height 256 --sh-degree 0 --steps 200 --warmup 20 --hip-graph
    c 300k --triangles 300000 --width 800 --height 800 --steps 40 --warmup 5
    c 2M --triangles 2000000 --steps 15 --warmup 4
    c 93k3d --triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --steps 100 --warmup 10
    c 93k3d_graph --triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --steps 100 --warmup 10 --hip-graph
    c 93k3d_g50 --triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D --gamma 50 --steps 100 --warmup 10
    c 5M3d --triangles 5000000 --sh-degree 0 --rasterizer 3D --steps 10 --warmup 3
    c 5M2d --triangles 5000000 --sh-degree 0 --steps 10 --warmup 3
    c head3d --rasterizer 3D --steps 20 --warmup 5
    c head_syncfree --sync-free --steps 20 --warmup 5
    c head_graph --hip-graph --steps 20 --warmup 5
    c head_centered --scene-mode centered --steps 20 --warmup 5
    c head_g50 --gamma 50 --steps 20 --warmup 5
    c head_adam --with-optimizer --steps 20 --warmup 5
    bench forward_only --forward-only --steps 40 --warmup 5
    bench forward_only --forward-only --triangles 300000 --width 800 --height 800 --steps 80 --warmup 10
    for cfg in headline lego300k mesh93k mesh93k_g50; do train $cfg; done
    loss; loss 800 800; loss 2160 3840
}

eval "$@"
