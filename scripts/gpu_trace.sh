#!/bin/bash
# Kernel timeline (start/end per dispatch) of one configuration: bash scripts/gpu_trace.sh tag "ENV=VAL ..."
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
# (the library destroys its pooled CU-masked streams at exit only when asked: rocprofv3 crashes on live ones)
for kv in $1; do export $kv; done
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$TAG -o t -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check ${BENCH_ARGS} > $OUT/trace_$TAG.log 2>&1
echo rc=$?
find $OUT/trace_$TAG -name "*kernel_trace.csv" -exec ls -la {} \;
