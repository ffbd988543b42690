#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc.sh for the roofline's
`traffic` field.

    python scripts/pmc_summary.py gpurun_out/pmc_<tag>_FETCH_SIZE gpurun_out/pmc_<tag>_WRITE_SIZE profiles/r01

HBM bytes per launch of the dominant kernel (k_update<128,...>) =
(2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 / launches: FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (the counter tallies
128-byte requests at 64 B), WRITE_SIZE is exact.  The algorithmic bytes come from
the program tables: every unit reads its A and B rows once ((M + N) * K * 8 B) and
read-modify-writes its destination entries (16 B each).
"""
import glob
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def counter_by_kernel(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    t = pd.read_csv(f)
    t = t[t["Counter_Name"] == name]
    g = t.groupby("Kernel_Name")["Counter_Value"].agg(["count", "sum"]).sort_values("sum", ascending=False)
    return g


def algorithmic_bytes(config="nd24k_like"):
    from spllt_amd import api, matgen
    A, order, cfg = matgen.build_config(config, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    L, units, tiles = f.program("launches"), f.program("units"), f.program("tiles")
    bw = f.sym("bcol_width")
    total, launches = 0.0, 0
    for l in L:
        if l[0] != 1 or l[4] != 128:
            continue
        launches += 1
        for uid in np.unique(tiles[l[2]:l[2] + l[3]]["unit"]):
            u = units[uid]
            M, N = float(u["M"]), float(u["N"])
            if u["nseg"] == 1:
                K = float(u["klen"]) if u["klen"] >= 0 else float(bw[u["src_bcol0"]])
            else:
                K = float(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
            dest = M * N - (0.5 * N * (N - 1) if u["lower"] else 0.0)
            total += (M + N) * K * 8 + 16 * dest
    return total, launches


def main():
    dfetch, dwrite, out = sys.argv[1], sys.argv[2], sys.argv[3]
    res = {}
    for d, name in ((dfetch, "FETCH_SIZE"), (dwrite, "WRITE_SIZE")):
        g = counter_by_kernel(d, name)
        with open(os.path.join(out, f"pmc_{name}_by_kernel.csv"), "w") as fh:
            fh.write("kernel,calls,sum_KB,avg_KB\n")
            for k, r in g.iterrows():
                fh.write(f"\"{k[:60]}\",{int(r['count'])},{r['sum']:.1f},{r['sum'] / r['count']:.2f}\n")
        sel = g[g.index.str.contains("k_update<128")]
        res[name] = {"launches_traced": int(sel["count"].sum()), "sum_KB": float(sel["sum"].sum()),
                     "avg_KB": float(sel["sum"].sum() / sel["count"].sum())}
    alg, nl = algorithmic_bytes()
    hbm = (2 * res["FETCH_SIZE"]["avg_KB"] + res["WRITE_SIZE"]["avg_KB"]) * 1024
    summary = {"kernel": "k_update<128,16,4,2>", "workload": "nd24k_like",
               "launches_per_factorization": nl,
               "algorithmic_bytes_per_factorization": alg,
               "algorithmic_bytes_per_launch": alg / nl, "pmc": res,
               "hbm_bytes_per_launch": hbm,
               "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at "
                       "64 B; calibrated there for 16-B/lane loads, this kernel issues 8-B/lane loads "
                       "so x2 is an upper bound); WRITE_SIZE exact. Units KB = 1024 B."}
    with open(os.path.join(out, "pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
