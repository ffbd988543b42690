#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc.sh for the roofline's
`traffic` field.

    python scripts/pmc_summary.py gpurun_out/pmc_<tag>_FETCH_SIZE gpurun_out/pmc_<tag>_WRITE_SIZE profiles/r01 \
        [gpurun_out/pmc_<tag>_MFMA_BUSY gpurun_out/pmc_<tag>_MFMA_MOPS]

HBM bytes per launch of each update kernel (k_update<128|64|32,...>) =
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


KERNELS = {128: "k_update<128, 16, 4, 2>", 64: "k_update<64, 16, 2, 2>", 32: "k_update<32, 32, 2, 2>"}


def algorithmic_bytes(config="nd24k_like"):
    """per tile size: (algorithmic bytes per factorization, launches per factorization)"""
    from spllt_amd import api, matgen
    A, order, cfg = matgen.build_config(config, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    L, units, tiles = f.program("launches"), f.program("units"), f.program("tiles")
    bw = f.sym("bcol_width")
    out = {T: [0.0, 0] for T in KERNELS}
    for l in L:
        if l[0] != 1 or int(l[4]) not in out:
            continue
        T = int(l[4])
        out[T][1] += 1
        tl = tiles[l[2]:l[2] + l[3]]
        for uid in np.unique(tl["unit"]):
            u = units[uid]
            # a unit may be split over two tile sizes (ragged tile columns): charge this
            # launch with the columns its tiles cover
            mine = tl[tl["unit"] == uid]
            ncols = min(float(u["N"]), float(len(np.unique(mine["tj"])) * T))
            M, N = float(u["M"]), ncols
            if u["nseg"] == 1:
                K = float(u["klen"]) if u["klen"] >= 0 else float(bw[u["src_bcol0"]])
            else:
                K = float(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
            if u["mode"] == 2:
                dest = 8 * M * N              # TRSM: X written once (operand A is the same block)
            else:
                dest = 16 * (M * N - (0.5 * N * (N - 1) if u["lower"] else 0.0))
            out[T][0] += (M + N) * K * 8 + dest
    return out


def mfma_by_kernel(dbusy, dmops):
    """MFMA pipe utilisation and executed matrix flops per update kernel.
    SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs; GRBM_GUI_ACTIVE comes back
    summed over the 8 XCDs (checked against the kernel durations), so
    util = busy / (gui_active / 8 * 1024 SIMDs).  MOPS_F64 * 512 = executed flops."""
    res = {}
    if dbusy:
        f = glob.glob(os.path.join(dbusy, "**", "*counter_collection.csv"), recursive=True)[0]
        t = pd.read_csv(f)
        g = t.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name",
                          values="Counter_Value", aggfunc="sum").reset_index()
        for T, kname in KERNELS.items():
            sel = g[g["Kernel_Name"].str.contains(kname, regex=False)]
            if len(sel):
                busy, act = sel["SQ_VALU_MFMA_BUSY_CYCLES"].sum(), sel["GRBM_GUI_ACTIVE"].sum()
                res.setdefault(kname, {})["mfma_util_percent"] = float(100 * busy / (act / 8 * 1024))
    if dmops:
        f = glob.glob(os.path.join(dmops, "**", "*counter_collection.csv"), recursive=True)[0]
        t = pd.read_csv(f)
        t = t[t["Counter_Name"] == "SQ_INSTS_VALU_MFMA_MOPS_F64"]
        for T, kname in KERNELS.items():
            sel = t[t["Kernel_Name"].str.contains(kname, regex=False)]
            if len(sel):
                res.setdefault(kname, {})["executed_mfma_flops_per_launch"] = float(
                    sel["Counter_Value"].sum() * 512 / sel["Dispatch_Id"].nunique())
    return res


def main():
    dfetch, dwrite, out = sys.argv[1], sys.argv[2], sys.argv[3]
    dbusy = sys.argv[4] if len(sys.argv) > 4 else None
    dmops = sys.argv[5] if len(sys.argv) > 5 else None
    per = {}
    for d, name in ((dfetch, "FETCH_SIZE"), (dwrite, "WRITE_SIZE")):
        g = counter_by_kernel(d, name)
        with open(os.path.join(out, f"pmc_{name}_by_kernel.csv"), "w") as fh:
            fh.write("kernel,calls,sum_KB,avg_KB\n")
            for k, r in g.iterrows():
                fh.write(f"\"{k[:60]}\",{int(r['count'])},{r['sum']:.1f},{r['sum'] / r['count']:.2f}\n")
        for T, kname in KERNELS.items():
            sel = g[g.index.str.contains(kname, regex=False)]
            if len(sel) == 0:
                continue
            per.setdefault(kname, {})[name] = {"launches_traced": int(sel["count"].sum()),
                                               "sum_KB": float(sel["sum"].sum()),
                                               "avg_KB": float(sel["sum"].sum() / sel["count"].sum())}
    alg = algorithmic_bytes()
    kernels = {}
    for T, kname in KERNELS.items():
        if kname not in per or "FETCH_SIZE" not in per[kname] or "WRITE_SIZE" not in per[kname]:
            continue
        a, nl = alg[T]
        kernels[kname] = {"launches_per_factorization": nl,
                          "algorithmic_bytes_per_launch": a / max(nl, 1),
                          "hbm_bytes_per_launch": (2 * per[kname]["FETCH_SIZE"]["avg_KB"] +
                                                   per[kname]["WRITE_SIZE"]["avg_KB"]) * 1024,
                          "pmc": per[kname]}
    for kname, v in mfma_by_kernel(dbusy, dmops).items():
        if kname in kernels:
            kernels[kname].update(v)
    summary = {"workload": "nd24k_like", "kernels": kernels,
               "note": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB * 1024 per launch: FETCH_SIZE doubled per "
                       "MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B; calibrated there for "
                       "16-B/lane loads, these kernels issue 8-B/lane loads so x2 is an upper bound); "
                       "WRITE_SIZE exact.  algorithmic bytes: every unit reads its A and B rows once and "
                       "read-modify-writes its destination entries (16 B each)."}
    with open(os.path.join(out, "pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
