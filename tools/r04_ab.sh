#!/bin/bash
# product vs variant libraries alternating on one box: all kernels_avg_ms of bench.py.  usage: tools/r04_ab.sh <lib.so>... [-- bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2; do
for L in "" "${LIBS[@]}"; do
  TS2D_LIBRARY_PATH=${L:+$R/$L} timeout 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_avg_ms']
print('${L:-product}'.split('/')[-1], j['ms_per_step'], ' '.join(f'{a}={b:.4f}' for a,b in k.items()))"
done; done
