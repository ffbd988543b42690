#!/bin/bash
# A/B of the subtree tasks (L_SUBTREE): bench workload at several budgets, then the small sizes
out=gpurun_out/sub; mkdir -p $out
for b in 0 300 1000 2000; do
  SPLLT_SUBTREE_US=$b python bench.py --no-extra-configs > $out/bench_$b.json 2> $out/bench_$b.err || exit 1
done
for b in 0 300 1000; do
  SPLLT_SUBTREE_US=$b python scripts/dbg/small_probe.py > $out/small_$b.txt 2>&1 || exit 1
done
