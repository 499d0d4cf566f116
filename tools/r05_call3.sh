#!/bin/bash
# Round 5, third lease: emission with the record prefetched under the scan (A/B against the previous commit's library), HIP-graph replay lines,
# the N > 1 bench path on one GPU over gloo with and without --sparse-exchange, the re-stated 3D spread tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_reference_gpu.py tests/test_async_forward_gpu.py tests/test_parity3d_gpu.py tests/test_parity_gpu.py tests/test_speculative_forward_gpu.py tests/test_multigpu_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|rc=|set aside" | head -40
echo "== 2D headline: product / previous alternating"; bash tools/ab_bench.sh $R/tools/bin/libts2d_prev.so --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tee $O/ab_2d.txt
echo "== 3D headline: product / previous alternating"; bash tools/ab_bench.sh $R/tools/bin/libts2d_prev.so --steps 20 --warmup 5 --rasterizer 3D 2>&1 | grep -v amdgpu.ids | tee $O/ab_3d.txt
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
S3="--triangles 93000 --width 1600 --height 1600 --sh-degree 0 --rasterizer 3D"
S0="--triangles 10000 --width 256 --height 256 --sh-degree 0"
for cfg in "$S3" "$S0" "--triangles 300000 --width 800 --height 800" ""; do
  $B $cfg --hip-graph 2>$O/err_graph.txt | tail -1 >> $O/graph.jsonl || { echo "graph failed: $cfg"; grep -v amdgpu.ids $O/err_graph.txt | tail -5; }
done
$B --triangles 5000000 --sh-degree 0 2>/dev/null | tail -1 >> $O/configs.jsonl
$B --triangles 5000000 --sh-degree 0 --rasterizer 3D 2>/dev/null | tail -1 >> $O/configs.jsonl
echo "== two ranks on one GPU over gloo (functional): dense / sparse exchange"
for X in "" "--sparse-exchange"; do
  TS2D_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --settle-steps 0 --triangles 200000 --width 800 --height 800 $X 2>$O/err_gloo.txt | tail -1 >> $O/gloo2.jsonl || grep -v amdgpu.ids $O/err_gloo.txt | tail -8
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.jsonl")):
    for l in open(f):
        try: j=json.loads(l)
        except Exception: print(f, "BAD", l[:300]); continue
        c=j["config"]; k=j.get("kernels_avg_ms",{})
        print(f.split("/")[-1], c["rasterizer"], c["triangles"], c["width"], c["forward"][:12], "ms", j["ms_per_step"], "host", c["host_step_ms"]["median"], "dev", c["device_step_ms"]["median"], "idle", c.get("gpu_idle_ms_per_step"), "exch", c.get("exchange"),
              " ".join(f"{a}={b:.4f}" for a,b in k.items()))
PY
