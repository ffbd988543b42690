#!/usr/bin/env python3
"""DESIGN.md section 2 said: "an engine created in a process that has freed other large
allocations before ran 2x slower".  Reproduce it in one script: the same factorization
(bench workload) in a fresh process, and in a process whose device heap was churned first
(large tensors allocated and freed in an order that leaves holes), with where the arena landed.

  python scripts/heap_cliff.py [config]            # runs both scenarios in child processes
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(config, scenario):
    import numpy as np
    import torch
    from spllt_amd import api, matgen
    torch.cuda.set_device(0)
    A, order, cfg = matgen.build_config(config, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    free0, total = torch.cuda.mem_get_info()
    if scenario != "fresh":
        # churn: tensors of 0.3 .. 6 GB, freed in an interleaved order, the cache emptied, twice
        rng = np.random.default_rng(1)
        for rounds in range(2):
            ts = [torch.empty(int(s * 2**30), dtype=torch.uint8, device="cuda")
                  for s in rng.uniform(0.3, 6.0, size=24)]
            for i in list(range(0, 24, 2)) + list(range(1, 24, 2)):
                ts[i] = None
                if i % 5 == 0:
                    torch.cuda.empty_cache()
            torch.cuda.empty_cache()
        if scenario == "churn+hold":
            hold = [torch.empty(int(1.3 * 2**30), dtype=torch.uint8, device="cuda") for _ in range(6)]
            hold = hold[::2]              # holes of 1.3 GB between kept blocks
            torch.cuda.empty_cache()
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    dval = torch.tensor(val, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        f.factor_dev(dval.data_ptr()).wait()
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        f.factor_dev(dval.data_ptr()).wait()
    ms = (time.perf_counter() - t0) / steps * 1e3
    p = f.device_factor_ptr() or 0
    free1, _ = torch.cuda.mem_get_info()
    print(json.dumps({"scenario": scenario, "ms_per_step": round(ms, 3), "arena_ptr": hex(p),
                      "arena_mod_2MiB": p % (2 << 20), "arena_mod_1GiB_MiB": (p % (1 << 30)) >> 20,
                      "free_before_GiB": round(free0 / 2**30, 2), "free_after_GiB": round(free1 / 2**30, 2)}))
    f.close()


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(sys.argv[1], sys.argv[2])
    else:
        config = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
        for sc in ("fresh", "churn", "churn+hold"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), config, sc], capture_output=True, text=True,
                               timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(lines[-1] if lines else f"{sc}: failed\n{r.stderr[-1500:]}", flush=True)
