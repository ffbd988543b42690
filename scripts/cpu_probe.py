#!/usr/bin/env python3
"""Host-side probe for the cpu_baseline leg: usable cores and oracle timing."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "OMP env",
      {k: v for k, v in os.environ.items() if k.startswith(("OMP", "GOMP", "MKL", "KMP"))}, flush=True)
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cgroup cpu.max n/a", e)
import numpy as np  # noqa: E402
from spllt_amd import api, matgen  # noqa: E402
from oracle import pyoracle  # noqa: E402

gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
A, order, cfg = matgen.build_config("nd24k_like", 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
if gpu:
    f.factor(val).wait()
    print("after GPU init: affinity", len(os.sched_getaffinity(0)), flush=True)
o = pyoracle.OracleFactor.from_factorization(f, variant="mkl")
for th in (16, 16, 8):
    t0 = time.time()
    o.factorize(val, th)
    print("threads", th, "seconds %.2f" % (time.time() - t0), flush=True)
