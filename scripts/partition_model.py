#!/usr/bin/env python3
"""What the subtree partition would deliver on w GPUs, measured on ONE device, one rank-engine
per process (a fresh device heap each time: engines that share a fragmented heap run slower):

    python scripts/partition_model.py <config> <width> <rank>      -> one JSON line

rank r of a width-w partition runs its own-subtree phase alone; rank 0 also runs the
replicated top-tree phase (on its own, unsummed exchange buffer: the top tree then misses the
other ranks' Schur complements, which changes the values but neither the structure nor the
positive definiteness, so the time is the same).  t(w) ~ max_r t_sub(r) + t_allreduce(volume)
+ t_top.  scripts/partition_model.sh loops over the ranks and prints the summary."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spllt_amd import api, matgen  # noqa: E402

cfg_name, w, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
A, order, cfg = matgen.build_config(cfg_name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
dval = torch.tensor(val, device="cuda")
f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=w > 1, ncpu=w, order=order)
xel = 0
if w > 1:
    xel = f.set_partition(r, w)
    xb = torch.zeros(max(xel, 1), dtype=torch.float64, device="cuda")
    f.set_exchange_buffer(xb.data_ptr())
tsub = ttop = None
for rep in range(2):   # second repetition is the measurement (the other ranks stop at the exchange point)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.factor_dev(dval.data_ptr())
    f.wait()
    tsub = (time.perf_counter() - t0) * 1e3
    if w > 1 and r == 0:     # only rank 0's buffer holds the top tree's own entries of A
        t0 = time.perf_counter()
        f.continue_after_exchange()
        f.wait()
        ttop = (time.perf_counter() - t0) * 1e3
si = f.sym_info()
print(json.dumps({"config": cfg_name, "width": w, "rank": r, "subtree_ms": round(tsub, 2),
                  "top_ms": None if ttop is None else round(ttop, 2), "exchange_MB": round(xel * 8 / 1e6, 1),
                  "flops_sym_G": round(si["flops"] / 1e9, 1)}), flush=True)
