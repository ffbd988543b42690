#!/bin/bash
# PMC passes for the roofline: HBM traffic (FETCH_SIZE, WRITE_SIZE) and matrix-core use
# (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_F64); separate runs,
# kernel-trace only.
set -o pipefail
TAG=${1:-r01}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in ${PMC_PASSES:-FETCH_SIZE WRITE_SIZE MFMA_BUSY MFMA_MOPS}; do
  case $C in
    MFMA_BUSY) CTRS="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    MFMA_MOPS) CTRS="SQ_INSTS_VALU_MFMA_MOPS_F64";;
    *) CTRS=$C;;
  esac
  timeout -k 10 400 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $OUT/pmc_${TAG}_$C.log 2>&1
  echo "$C rc=$?"
  find $OUT/pmc_${TAG}_$C -name "*.csv" | head
done
