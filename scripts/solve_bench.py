#!/usr/bin/env python3
"""Times spllt_solve on the device-resident factor of a bench configuration."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spllt_amd import api, matgen  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A, order, cfg = matgen.build_config(cfg_name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
f.factor(val).wait()
X = np.ones((n, nrhs)) if nrhs > 1 else np.ones(n)
B = A @ X
x = f.solve(B)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    x = f.solve(B)
    ts.append(time.perf_counter() - t0)
r = B - A @ x
print(f"{cfg_name}: n={n} nrhs={nrhs} solve min {min(ts) * 1e3:.2f} ms  median {sorted(ts)[2] * 1e3:.2f} ms  "
      f"resid {np.linalg.norm(r) / np.linalg.norm(B):.2e}")
