#!/bin/bash
# bash scripts/partition_model.sh <config> "<widths>" [dist]   (on the GPU box, from the repository root)
# dist: the top tree distributed over the ranks (every rank runs its share), else replicated
CFG=$1; WIDTHS=${2:-"2 4 8"}; MODE=${3:-}
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/partition_model_${CFG}${MODE:+_$MODE}.txt
: > $OUT
for w in $WIDTHS; do
  for ((r = 0; r < w; r++)); do
    timeout -k 10 300 python scripts/partition_model.py $CFG $w $r $MODE 2>/dev/null | tail -1 >> $OUT || echo "{\"config\": \"$CFG\", \"width\": $w, \"rank\": $r, \"failed\": true}" >> $OUT
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
        r0 = rs[0]
        line = (f"# {r0['config']} width {w}, top tree {r0['top']}: own-subtree phase max {max(sub):.1f} ms (min {min(sub):.1f}), "
                f"top-tree phase per rank max {(max(top) if top else 0.0):.1f} / sum {sum(top):.1f} ms, "
                f"exchanges {r0['exchanges']}: all-reduce {r0['allreduce_MB']:.0f} MB, reduce-scatter {r0['reduce_scatter_MB']:.0f} MB, "
                f"broadcasts {r0['broadcast_MB']:.0f} MB; sum without communication {max(sub) + (max(top) if top else 0.0):.1f} ms, "
                f"F_sym {r0['flops_sym_G']:.0f} GFLOP")
        print(line)
        fh.write(line + "\n")
PY
