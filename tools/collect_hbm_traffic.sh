#!/bin/bash
# HBM traffic of every kernel of one bench step from the TCC fabric counters (MI355X_MICROARCH.md "HBM" section):
# separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), values in KiB;
# gfx950 correction: FETCH_SIZE under-reports wide (16 B/lane) reads by exactly 2x -> doubled; WRITE_SIZE is calibrated on
# the 64 000 000-byte clear of the gradient records (zero_words4_kernel, 1 M triangles x 64 B) that is part of every step; the summary script EXITS NON-ZERO
# when that kernel reports nothing (round 5's file carried a silent default of 1.0).  Output: gpurun_out/hbm_traffic.json (copy to profiles/).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmc_$C.log 2>&1
done
# calibration kernels of known size under the SAME counters (tools/gather_calib.hip: 64-byte record gather / streaming read; streaming store / record scatter)
mkdir -p $R/tools/bin && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/gather_calib.hip -o $R/tools/bin/gather_calib
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_calib_$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_calib_$C -- $R/tools/bin/gather_calib > $R/gpurun_out/pmc_calib_$C.log 2>&1
done
python $R/tools/hbm_traffic_summary.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE 64000000 $R/gpurun_out/pmc_calib_FETCH_SIZE $R/gpurun_out/pmc_calib_WRITE_SIZE > $R/gpurun_out/hbm_traffic.json
cat $R/gpurun_out/hbm_traffic.json
