#!/bin/bash
# bash scripts/partition_model.sh <config> "<widths>"   (on the GPU box, from the repository root)
CFG=$1; WIDTHS=${2:-"2 4 8"}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/partition_model_$CFG.txt
: > $OUT
for w in 1 $WIDTHS; do
  for ((r = 0; r < w; r++)); do
    timeout -k 10 300 python scripts/partition_model.py $CFG $w $r 2>/dev/null | tail -1 >> $OUT || echo "{\"config\": \"$CFG\", \"width\": $w, \"rank\": $r, \"failed\": true}" >> $OUT
    tail -1 $OUT
  done
done
python - "$OUT" <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
by = collections.defaultdict(list)
for r in rows:
    if not r.get("failed"):
        by[r["width"]].append(r)
with open(sys.argv[1], "a") as fh:
    for w, rs in sorted(by.items()):
        sub = [r["subtree_ms"] for r in rs]
        top = [r["top_ms"] for r in rs if r["top_ms"] is not None]
        line = (f"# {rs[0]['config']} width {w}: own-subtree phase max {max(sub):.1f} ms (min {min(sub):.1f}), top tree "
                f"{(max(top) if top else 0.0):.1f} ms, exchange {rs[0]['exchange_MB']:.0f} MB, sum without exchange "
                f"{max(sub) + (max(top) if top else 0.0):.1f} ms, F_sym {rs[0]['flops_sym_G']:.0f} GFLOP")
        print(line)
        fh.write(line + "\n")
PY
