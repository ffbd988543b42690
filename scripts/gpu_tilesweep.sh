#!/bin/bash
# A/B of the tile-size thresholds of the update launches (nd24k_like, default engine).
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $OUT
IFS=";" read -ra CFGS <<< "${SWEEP:-1024 2048;2048 2048;2048 4096;4096 4096;4096 8192;100000 4096;100000 100000}"
for cfg in "${CFGS[@]}"; do
  IFS=" "
  set -- $cfg
  SPLLT_TILE_SMALL=$1 SPLLT_TILE_TINY=$2 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check > $OUT/ts_$1_$2.json 2> $OUT/ts_$1_$2.err || exit 1
  python - "$1" "$2" $OUT/ts_$1_$2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("small",sys.argv[1],"tiny",sys.argv[2],"ms",d["ms_per_step"],"GF",d["value"],flush=True)
PY
done
