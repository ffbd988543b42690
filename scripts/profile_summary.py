#!/usr/bin/env python3
"""Merges the rocprofv3 outputs of scripts/gpu_profile.sh with the program tables:

    python scripts/profile_summary.py <tag> <dir> "<bench.py args>"

For ONE factorization of the traced run (the second: the first timed step) every kernel
dispatch is matched with the launch of the exported program it belongs to (dispatches are
issued in program order), which gives per launch category -- chain / trsm / inpanel / next /
trailing / between -- the device time, the algorithmic flops, HBM bytes from the PMC passes
(2 * FETCH_SIZE + WRITE_SIZE: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
gfx950, WRITE_SIZE exact incl. fp64 atomics) and the matrix-pipe utilisation.  For the
inter-node updates (`between`, the scatter-add epilogue) it also reports the achieved atomic
GB/s = 8 B x destination entries / kernel time (north_star: "achieved HBM GB/s on scatter-add").
Writes <dir>/summary.json, <dir>/categories.csv, <dir>/kernel_stats.csv (copy of rocprof's).
"""
import glob
import json
import os
import shutil
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_TFLOPS, PEAK_HBM_TBS, ATOMIC_TBS = 78.6, 8.0, 1.3


def program_tables(args):
    import argparse
    from spllt_amd import api, matgen
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="nd24k_like")
    ap.add_argument("--ordering", default="geometric")
    a, _ = ap.parse_known_args(args.split())
    A, order, cfg = matgen.build_config(a.config, 1.0)
    if a.ordering == "builtin":
        order = None
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order,
                          engine_flags=int(os.environ.get("SPLLT_ENGINE_FLAGS", "0")))
    return a.config, f


def categorize(f):
    L, units, tiles = f.program("launches"), f.program("units"), f.program("tiles")
    bc_off, bw = f.sym("bcol_off"), f.sym("bcol_width")
    rows = []
    for l in L:
        if l[3] <= 0 or l[0] == 2:
            continue
        kind, first, count, T = int(l[0]), int(l[2]), int(l[3]), int(l[4])
        rec = dict(kind=kind, tile=T, flops=float(l[5]), entries=0.0, alg_bytes=0.0, cat="other")
        if kind in (4, 8):
            rec["cat"] = "chain"
        elif kind == 9:
            rec["cat"] = "trsm"
            for uid in np.unique(tiles[first:first + count]["unit"]):
                u = units[uid]
                M, N = float(u["M"]), float(u["N"])
                rec["entries"] += M * N
                rec["alg_bytes"] += M * N * 16 + N * N * 8
        elif kind == 7:
            rec["cat"] = "panel"      # fused panel step (k_panel)
        elif kind == 6:
            rec["cat"] = "gather"
        elif kind == 1:
            tl = tiles[first:first + count]
            u0 = units[int(tl[0]["unit"])]
            if u0["mode"] == 2:
                rec["cat"] = "trsm"
            elif u0["mode"] in (1, 3):
                rec["cat"] = "between"
            elif bc_off[int(u0["src_bcol0"])] == u0["d_off"]:
                rec["cat"] = "inpanel"
            else:
                rec["cat"] = {0: "next", 3: "next_rest", 1: "trailing"}.get(int(l[6]), "update")
            for uid in np.unique(tl["unit"]):
                u = units[uid]
                mine = tl[tl["unit"] == uid]
                N = min(float(u["N"]), float(len(np.unique(mine["tj"])) * T))
                M = float(u["M"])
                K = (float(u["klen"]) if u["klen"] >= 0 else float(bw[u["src_bcol0"]])) if u["nseg"] == 1 else \
                    float(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
                ent = M * N - (0.5 * N * (N - 1) if u["lower"] else 0.0)
                rec["entries"] += ent
                # operands once + destination: TRSM / buffer 8 B (store), atomic 8 B, RMW 16 B
                dest = 8 * ent if (u["mode"] in (1, 2, 3) or u["atomic"]) else 16 * ent
                rec["alg_bytes"] += (M + N) * K * 8 + dest
        rows.append(rec)
    return pd.DataFrame(rows)


def one_factorization(trace_csv, which=1):
    d = pd.read_csv(trace_csv)
    sv = d.index[d["Kernel_Name"].str.contains("k_scatter_val")].tolist()
    lo, hi = sv[which], (sv[which + 1] if which + 1 < len(sv) else len(d))
    f = d.iloc[lo + 1:hi].copy()
    f = f[f["Kernel_Name"].str.contains("spx::")]
    f = f.sort_values("Dispatch_Id")
    f["us"] = (f["End_Timestamp"] - f["Start_Timestamp"]) / 1e3
    return f, (f["End_Timestamp"].max() - d.iloc[lo]["Start_Timestamp"]) / 1e6


def counters(dirname, which=1):
    fs = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return None
    t = pd.read_csv(fs[0])
    g = t.pivot_table(index=["Dispatch_Id", "Kernel_Name"], columns="Counter_Name", values="Counter_Value",
                      aggfunc="sum").reset_index().sort_values("Dispatch_Id")
    sv = g.index[g["Kernel_Name"].str.contains("k_scatter_val")].tolist()
    pos = [g.index.get_loc(i) for i in sv]
    lo, hi = pos[which], (pos[which + 1] if which + 1 < len(pos) else len(g))
    f = g.iloc[lo + 1:hi]
    return f[f["Kernel_Name"].str.contains("spx::")].reset_index(drop=True)


def main():
    tag, d, args = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    config, f = program_tables(args)
    prog = categorize(f)
    trace = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
    disp, span_ms = one_factorization(trace)
    assert len(disp) == len(prog), (len(disp), len(prog))
    prog["us"] = disp["us"].values
    prog["kernel"] = disp["Kernel_Name"].str.replace("void spx::", "").str.replace("spx::", "").str.slice(0, 28).values
    for name, col in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb")):
        c = counters(os.path.join(d, "pmc_" + name))
        prog[col] = c[name].values if c is not None and len(c) == len(prog) else np.nan
    c = counters(os.path.join(d, "pmc_MFMA_BUSY"))
    if c is not None and len(c) == len(prog):
        prog["mfma_busy"], prog["gui_active"] = c["SQ_VALU_MFMA_BUSY_CYCLES"].values, c["GRBM_GUI_ACTIVE"].values
    else:
        prog["mfma_busy"] = prog["gui_active"] = np.nan
    c = counters(os.path.join(d, "pmc_L2"))
    if c is not None and len(c) == len(prog) and "TCC_HIT_sum" in c:
        prog["l2_hit"], prog["l2_miss"] = c["TCC_HIT_sum"].values, c["TCC_MISS_sum"].values
    else:
        prog["l2_hit"] = prog["l2_miss"] = np.nan
    prog["hbm_bytes"] = (2 * prog["fetch_kb"] + prog["write_kb"]) * 1024
    rows = []
    for key, g in list(prog.groupby("cat")) + list(prog[prog["kind"] == 1].groupby("kernel")):
        t = g["us"].sum() * 1e-6
        rows.append({"group": key, "launches": len(g), "ms": round(t * 1e3, 3), "gflop": round(g["flops"].sum() / 1e9, 2),
                     "tflops": round(g["flops"].sum() / t / 1e12, 2) if t > 0 else 0.0,
                     "frac_mfma_peak": round(g["flops"].sum() / t / 1e12 / PEAK_TFLOPS, 3) if t > 0 else 0.0,
                     "mfma_pipe_busy_pct": round(float(100 * g["mfma_busy"].sum() / (g["gui_active"].sum() / 8 * 1024)), 1)
                     if g["gui_active"].sum() > 0 else None,
                     "alg_GB": round(g["alg_bytes"].sum() / 1e9, 3), "hbm_GB": round(g["hbm_bytes"].sum() / 1e9, 3),
                     "hbm_over_alg": round(g["hbm_bytes"].sum() / g["alg_bytes"].sum(), 2) if g["alg_bytes"].sum() > 0 else None,
                     "hbm_GBps": round(g["hbm_bytes"].sum() / t / 1e9, 1) if t > 0 else 0.0,
                     "l2_hit_pct": round(float(100 * g["l2_hit"].sum() / (g["l2_hit"].sum() + g["l2_miss"].sum())), 1)
                     if (g["l2_hit"].sum() + g["l2_miss"].sum()) > 0 else None,
                     "scatter_entries_M": round(g["entries"].sum() / 1e6, 1) if key == "between" else None,
                     "scatter_atomic_GBps": round(8 * g["entries"].sum() / t / 1e9, 1) if key == "between" and t > 0 else None})
    table = pd.DataFrame(rows)
    table.to_csv(os.path.join(d, "categories.csv"), index=False)
    si = f.sym_info()
    dom = prog[prog["kind"] == 1].groupby("kernel")["us"].sum().idxmax()
    gd = prog[(prog["kind"] == 1) & (prog["kernel"] == dom)]
    summary = {"tag": tag, "workload": config, "bench_args": args, "flops_sym": float(si["flops"]),
               "factorization_span_ms_traced": round(float(span_ms), 3),
               "sum_of_kernel_ms": round(float(prog["us"].sum() / 1e3), 3),
               "dominant_kernel": dom, "dominant_launches": int(len(gd)),
               "dominant_avg_launch_us": round(float(gd["us"].mean()), 2),
               "dominant_tflops_in_program": round(float(gd["flops"].sum() / gd["us"].sum() / 1e6), 2),
               "peaks": {"fp64_mfma_TFLOPs": PEAK_TFLOPS, "hbm_TBs": PEAK_HBM_TBS, "fp64_atomic_TBs": ATOMIC_TBS},
               "groups": rows,
               "kernels": {k: {"launches_per_factorization": int(len(g)),
                               "hbm_bytes_per_launch": float(g["hbm_bytes"].mean()),
                               "algorithmic_bytes_per_launch": float(g["alg_bytes"].mean()),
                               "mfma_util_percent": float(100 * g["mfma_busy"].sum() / (g["gui_active"].sum() / 8 * 1024))
                               if g["gui_active"].sum() > 0 else None}
                           for k, g in prog[prog["kind"] == 1].groupby("kernel")},
               "note": "one factorization of the traced run; hbm = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (FETCH doubled "
                       "per MI355X_MICROARCH.md, upper bound for these 8-B/lane loads); scatter_atomic_GBps = 8 B x "
                       "destination entries / time of the inter-node update launches (chip limit ~1300 GB/s)"}
    # kernel names as bench.py spells them
    summary["kernels"] = {("k_update<" + k.split("k_update<")[1].split(">")[0] + ">") if "k_update<" in k else
                          ("k_update_dma128" if "k_update_dma128" in k else k): v
                          for k, v in summary["kernels"].items()}
    with open(os.path.join(d, "summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    st = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(d, "kernel_stats.csv"))
    print(table.to_string())
    print(json.dumps({k: summary[k] for k in ("workload", "factorization_span_ms_traced", "dominant_kernel",
                                              "dominant_avg_launch_us", "dominant_tflops_in_program")}))


if __name__ == "__main__":
    main()
