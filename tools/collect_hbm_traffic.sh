#!/bin/bash
# HBM traffic of every kernel of one bench step from the TCC fabric counters (MI355X_MICROARCH.md "HBM" section):
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), values in KiB;
# gfx950 correction: FETCH_SIZE under-reports wide (16 B/lane) reads by exactly 2x -> doubled; WRITE_SIZE is calibrated on
# the 64 000 000-byte hipMemsetAsync of the gradient records (1 M triangles x 64 B) that is part of every step.  Output: gpurun_out/hbm_traffic.json (copy to profiles/).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_$C.log 2>&1
done
python $R/tools/hbm_traffic_summary.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE 64000000 > $R/gpurun_out/hbm_traffic.json
cat $R/gpurun_out/hbm_traffic.json
