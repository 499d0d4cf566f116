#!/bin/bash
# SPI resource-arbiter counters (why workgroups could not be placed) for every kernel of the step, TWO counters per pass (more: "exceeds the capabilities of the
# hardware", and rocprofv3 then hangs until killed); workload: bench.py.   usage: collect_spi_pmc.sh <outdir> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN" "SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_BAR_CU_FULL_CSN" "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN" "SPI_CSN_BUSY SPI_RA_TGLIM_CU_FULL_CSN"; do
  i=$((i+1)); rm -rf ${OUT}_$i
  timeout -s KILL 60 rocprofv3 --kernel-trace --pmc $P --output-format csv -d ${OUT}_$i -- python $R/bench.py --steps 3 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-kernel-events "$@" > ${OUT}_$i.log 2>&1
  echo "pass $i rc=$?"
  find ${OUT}_$i -name "*kernel_trace.csv" -delete 2>/dev/null
done
