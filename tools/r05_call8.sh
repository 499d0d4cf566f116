#!/bin/bash
# Round 5, eighth lease: the tests call 7 ran against a stale lab library, the ranged backward of the 3D variant, gloo-2 exchange variants.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_qmask_gpu.py tests/test_multigpu_gpu.py tests/test_optim_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v amdgpu.ids $O/pytest.log | grep -E "^FAILED|^ERROR|^E  |passed|failed|rc=" | head -30
for X in "" "--range-exchange 4" "--sparse-exchange"; do
  TS2D_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 20 --warmup 5 --settle-steps 0 --triangles 200000 --width 800 --height 800 $X 2>$O/err_gloo.txt | tail -1 >> $O/gloo2.jsonl || grep -v amdgpu.ids $O/err_gloo.txt | tail -8
done
python - <<PY
import json
for l in open("$O/gloo2.jsonl"):
    j=json.loads(l); print(j["ms_per_step"], j["config"]["exchange"])
PY
