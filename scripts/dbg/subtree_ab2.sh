#!/bin/bash
# small budgets of the subtree tasks on the small sizes
out=gpurun_out/sub; mkdir -p $out
for b in 40 100; do
  SPLLT_SUBTREES=1 SPLLT_SUBTREE_US=$b python scripts/dbg/small_probe.py > $out/small_$b.txt 2>&1 || exit 1
done
