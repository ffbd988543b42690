#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.
# Usage (from the repo root on the box): bash scripts/gpu_round.sh [tag]
set -o pipefail
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-.}
echo "== pytest -m gpu" | tee $OUT/status_$TAG.txt
timeout -k 10 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -5 $OUT/pytest_gpu_$TAG.log
if [ $rc -gt 1 ]; then echo "pytest died (rc=$rc): stopping"; exit $rc; fi
echo "== smoke" | tee -a $OUT/status_$TAG.txt
timeout -k 10 180 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1
rc=$?; echo "smoke rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -3 $OUT/smoke_$TAG.log
if [ $rc -ne 0 ]; then exit $rc; fi
echo "== bench" | tee -a $OUT/status_$TAG.txt
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --profile-out $OUT/launch_table_$TAG.txt > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
rc=$?; echo "bench rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -c 3000 $OUT/bench_$TAG.json; tail -5 $OUT/bench_$TAG.err
if [ $rc -ne 0 ]; then exit $rc; fi
echo "== rocprofv3 kernel trace" | tee -a $OUT/status_$TAG.txt
export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check > $OUT/rocprof_$TAG.log 2>&1)
rc=$?; echo "rocprof rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -3 $OUT/rocprof_$TAG.log
find $OUT/prof_$TAG -name "*stats*" | head
exit 0
