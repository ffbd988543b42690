#!/bin/bash
# Which clock and power does the chip hold while a configuration factorizes?  rocm-smi sampled
# beside a bench run:  bash scripts/clock_probe.sh <config> [steps]
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
CFG=${1:-serena_like}; STEPS=${2:-30}
( timeout -k 10 300 python bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-check --no-extra-configs > $OUT/clock_bench_$CFG.json 2> $OUT/clock_bench_$CFG.err ) &
BP=$!
: > $OUT/clock_samples_$CFG.txt
while kill -0 $BP 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr '\n' ' ' >> $OUT/clock_samples_$CFG.txt
  echo >> $OUT/clock_samples_$CFG.txt
  sleep 0.5
done
wait $BP; echo "bench rc=$?"
tail -c 400 $OUT/clock_bench_$CFG.json | head -c 400; echo
wc -l $OUT/clock_samples_$CFG.txt; head -3 $OUT/clock_samples_$CFG.txt; echo ...; tail -12 $OUT/clock_samples_$CFG.txt
