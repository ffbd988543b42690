#!/bin/bash
# A/B of engine variants on the bench workload (one process per variant; quick)
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check > $OUT/var_$name.json 2> $OUT/var_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/var_$name.json")); print("$name", d["value"], "GFLOP/s", d["ms_per_step"], "ms", d["roofline"]["achieved"])
except Exception as e:
    print("$name FAILED", e)
PY
}
run default SPLLT_ENGINE_FLAGS=0
run chain_strip SPLLT_ENGINE_FLAGS=4
run strip_only SPLLT_ENGINE_FLAGS=12
run single_stream SPLLT_ENGINE_FLAGS=2
