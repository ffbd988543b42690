#!/bin/bash
# Round profile of ONE configuration on the GPU box (from the repository root):
#   bash scripts/gpu_profile.sh <tag> "<bench.py args>"
# 1. rocprofv3 --kernel-trace --stats of the default bench command
# 2. separate rocprofv3 --pmc passes (kernel-trace only, as the pool requires): FETCH_SIZE,
#    WRITE_SIZE (HBM traffic), SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (matrix-pipe use),
#    TCC_HIT_sum + TCC_MISS_sum (L2 hit rate)
# Outputs under gpurun_out/prof_<tag>/; scripts/profile_summary.py turns them into profiles/.
set -o pipefail
TAG=$1; ARGS=$2
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof_$TAG
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
export TMPDIR=/tmp
# (the library destroys its pooled CU-masked streams at exit only when asked: rocprofv3 crashes on live ones)
cd /tmp
B="--steps 2 --warmup 1 --no-cpu-baseline --no-check --no-extra-configs $ARGS"
FAILED=0
# a pass whose process aborts at exit (round 2: SIGSEGV in __cxa_finalize with pooled CU-masked
# streams still alive when the runtime unloaded) still leaves its CSVs behind: say so, loudly
check_pass() {   # name, rc, log
  if [ "$2" -ne 0 ] || grep -q -e "dumped core" -e "Segmentation fault" -e "Aborted" "$3"; then
    echo "PASS $1 ABORTED (rc=$2): see $3"; FAILED=1
  else
    echo "$1 rc=0 (clean exit)"
  fi
}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py $B > $OUT/trace.log 2>&1
check_pass trace $? $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE MFMA_BUSY L2; do
  case $C in
    MFMA_BUSY) CTRS="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
    L2) CTRS="TCC_HIT_sum TCC_MISS_sum";;     # L2 hit rate per dispatch (MI355X_MICROARCH.md, L2)
    *) CTRS=$C;;
  esac
  timeout -k 10 500 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $REPO/bench.py $B > $OUT/pmc_$C.log 2>&1
  check_pass $C $? $OUT/pmc_$C.log
done
# keep what the summary needs, drop the rest (64 MiB merge limit)
find $OUT -name "*.csv" ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
python3 $REPO/scripts/profile_summary.py $TAG $OUT "$ARGS" > $OUT/summary.log 2>&1; echo "summary rc=$?"; tail -3 $OUT/summary.log
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT
[ $FAILED -eq 0 ] || { echo "at least one profiler pass did not exit cleanly"; exit 3; }
