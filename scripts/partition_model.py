#!/usr/bin/env python3
"""What the subtree partition would deliver on w GPUs, measured on ONE device, one rank-engine
per process (a fresh device heap each time: engines that share a fragmented heap run slower):

    python scripts/partition_model.py <config> <width> <rank> [dist]      -> one JSON line

rank r of a width-w partition runs its program alone: the own-subtree phase, then the top-tree
phase with the collectives left out (the exchange buffer then holds only this rank's own data:
the top tree misses the other ranks' contributions, which changes the values but neither the
structure nor the work, so the time is the same).
  replicated top tree (default): only rank 0 runs the top phase (every rank would do the same);
      t(w) ~ max_r t_sub(r) + t_allreduce(volume) + t_top
  dist: every rank runs its share of the distributed top tree (the panel chains of the block
      columns it owns and the updates of the destinations it owns);
      t(w) ~ max_r t_sub(r) + t_reduce_scatter + max(max_r t_top(r), chain of the top tree) + broadcasts
scripts/partition_model.sh loops over the ranks and prints the summary."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spllt_amd import api, matgen  # noqa: E402

cfg_name, w, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dist = len(sys.argv) > 4 and sys.argv[4] == "dist"
A, order, cfg = matgen.build_config(cfg_name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
dval = torch.tensor(val, device="cuda")
f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=w > 1, ncpu=w, order=order,
                      engine_flags=(8192 if dist else 16384) if w > 1 else 0)
xel = 0
if w > 1:
    xel = f.set_partition(r, w)
    xb = torch.zeros(max(xel, 1), dtype=torch.float64, device="cuda")
    f.set_exchange_buffer(xb.data_ptr())
tsub = ttop = None
nx = 0
for rep in range(2):   # second repetition is the measurement
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f.factor_dev(dval.data_ptr())
    f.wait()
    tsub = (time.perf_counter() - t0) * 1e3
    if w > 1 and (dist or r == 0):     # replicated: only rank 0's buffer holds the top tree's own entries of A
        t0 = time.perf_counter()
        nx = 0
        while f.pending_exchange() >= 0:
            f.continue_after_exchange()      # (no collective: see above)
            nx += 1
        try:
            f.wait()
        except api.SplltError:
            pass                             # incomplete sums: a pivot may fail, the work is the same
        ttop = (time.perf_counter() - t0) * 1e3
ex = f.program("exchanges") if w > 1 else []
vol = {int(k): 0 for k in (0, 1, 2)}
xit = f.program("xitems") if w > 1 else []
for kind, first, nit, elems, chunk in (ex.tolist() if w > 1 else []):
    # (elems is the END of the exchange's region in the buffer: the volume is counted from the items)
    if kind == 0:
        vol[0] += elems
    elif kind == 1:
        vol[1] += chunk * w
    elif kind == 2:
        vol[2] += int(xit[first:first + nit, 3].sum())
si = f.sym_info()
print(json.dumps({"config": cfg_name, "width": w, "rank": r, "top": "distributed" if dist else "replicated",
                  "subtree_ms": round(tsub, 2), "top_ms": None if ttop is None else round(ttop, 2),
                  "exchanges": nx, "allreduce_MB": round(vol[0] * 8 / 1e6, 1),
                  "reduce_scatter_MB": round(vol[1] * 8 / 1e6, 1), "broadcast_MB": round(vol[2] * 8 / 1e6, 1),
                  "flops_sym_G": round(si["flops"] / 1e9, 1)}), flush=True)
