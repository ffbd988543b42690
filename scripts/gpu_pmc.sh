#!/bin/bash
# PMC passes (HBM traffic) for the roofline: separate runs per counter, kernel-trace only.
set -o pipefail
TAG=${1:-r01}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$C -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check > $OUT/pmc_${TAG}_$C.log 2>&1
  echo "$C rc=$?"
  find $OUT/pmc_${TAG}_$C -name "*.csv" | head
done
