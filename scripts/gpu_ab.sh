#!/bin/bash
# Generic A/B: each argument is "ENV=VAL ENV=VAL ..." applied to one bench.py run.
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-check ${BENCH_ARGS} > $OUT/ab_$i.json 2> $OUT/ab_$i.err || { tail -3 $OUT/ab_$i.err; exit 1; }
  python - "$cfg" $OUT/ab_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-60s ms %.3f  GF %.0f  roofline kernel %.1f TF  host submit %.2f ms" % (sys.argv[1], d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["detail"].get("host_submit_ms_per_step", -1)), flush=True)
PY
done
