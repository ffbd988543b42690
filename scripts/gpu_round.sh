#!/bin/bash
# One GPU-box session: parity tests (ONE run: a failure is worked from its record, never re-run
# to see it again), smoke, bench, stand-alone probes.
# Usage (from the repo root on the box): bash scripts/gpu_round.sh [tag]
set -o pipefail
TAG=${1:-r02}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-.}
for i in 1; do
  echo "== pytest -m gpu (run $i)" | tee -a $OUT/status_$TAG.txt
  # (SPLLT_HIP_CRUMBS: the library's last step, should a call never return)
  SPLLT_HIP_CRUMBS=$OUT/crumbs_${TAG}_$i.txt timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 240 > $OUT/pytest_gpu_${TAG}_$i.log 2>&1
  rc=$?; echo "pytest rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -3 $OUT/pytest_gpu_${TAG}_$i.log
  if [ $rc -ne 0 ]; then echo "pytest failed (rc=$rc): stopping"; exit $rc; fi
done
echo "== smoke" | tee -a $OUT/status_$TAG.txt
timeout -k 10 180 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$TAG.log 2>&1
rc=$?; echo "smoke rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -2 $OUT/smoke_$TAG.log
[ $rc -ne 0 ] && exit $rc
echo "== bench" | tee -a $OUT/status_$TAG.txt
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --profile-out $OUT/launch_table_$TAG.txt > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
rc=$?; echo "bench rc=$rc" | tee -a $OUT/status_$TAG.txt; tail -c 1500 $OUT/bench_$TAG.json; tail -2 $OUT/bench_$TAG.err
[ $rc -ne 0 ] && exit $rc
if [ -n "$WITH_FLAN" ]; then
  timeout -k 10 600 python bench.py --steps 2 --warmup 1 --config flan_like --no-cpu-baseline --no-extra-configs > $OUT/bench_flan_$TAG.json 2> $OUT/bench_flan_$TAG.err; echo "flan rc=$?"; tail -c 600 $OUT/bench_flan_$TAG.json
fi
[ -x bin_tmp/cumask_probe ] && ./bin_tmp/cumask_probe > $OUT/cumask_probe_$TAG.txt 2>&1
for v in 16_4_2 32_4_2 16_2_4 16_2_2; do [ -x bin_tmp/ub_$v ] && { echo "== 128-tile BK_WM_WN=$v"; ./bin_tmp/ub_$v 8192 8192 | grep "T=128"; } ; done > $OUT/update_bench_128_variants_$TAG.txt 2>&1
exit 0
