#!/usr/bin/env python3
"""What the subtree partition would deliver on w GPUs, measured on ONE device:
every rank-engine of a width-w partition runs its own-subtree phase and the
replicated top-tree phase by itself; the exchange is a torch sum and only its
volume is reported.  t(w) ~ max_r t_sub(r) + t_allreduce(volume) + t_top."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spllt_amd import api, matgen  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
A, order, cfg = matgen.build_config(cfg_name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
dval = torch.tensor(val, device="cuda")
for w in (1, 2, 4, 8):
    fs, bufs = [], []
    for r in range(w):
        f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=w > 1, ncpu=w, order=order)
        if w > 1:
            xel = f.set_partition(r, w)
            xb = torch.zeros(max(xel, 1), dtype=torch.float64, device="cuda")
            f.set_exchange_buffer(xb.data_ptr())
            bufs.append(xb)
        fs.append(f)
    tsub, ttop = [], []
    for rep in range(2):
        tsub, ttop = [], []
        for f in fs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f.factor_dev(dval.data_ptr())
            f.wait()
            tsub.append((time.perf_counter() - t0) * 1e3)
        if w > 1:
            total = torch.stack(bufs).sum(dim=0)
            for xb in bufs:
                xb.copy_(total)
            torch.cuda.synchronize()
            for f in fs:
                t0 = time.perf_counter()
                f.continue_after_exchange()
                f.wait()
                ttop.append((time.perf_counter() - t0) * 1e3)
    flops = fs[0].sym_info()["flops"]
    xmb = bufs[0].numel() * 8 / 1e6 if w > 1 else 0.0
    print(f"{cfg_name} width {w}: subtree phase max {max(tsub):.2f} ms (min {min(tsub):.2f}), "
          f"top tree {max(ttop) if ttop else 0.0:.2f} ms, exchange {xmb:.0f} MB, "
          f"sum without exchange {max(tsub) + (max(ttop) if ttop else 0.0):.2f} ms, F_sym {flops / 1e9:.0f} GF", flush=True)
    print("   per rank: subtrees", [round(t, 2) for t in tsub], "top", [round(t, 2) for t in ttop], flush=True)
    for f in fs:
        f.close()
    del fs, bufs
