# A/B of two builds of libts2d.so on ONE box: bench.py's per-stage HIP-event averages, alternating.  usage: ab_bench.sh <other.so> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OTHER=$1; shift
for i in 1 2; do
  for L in "" "$OTHER"; do
    TS2D_LIBRARY_PATH=$L timeout 200 python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('${L:-product}'.split('/')[-1], j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items()))"
  done
done
