#!/usr/bin/env python3
"""Where the time of one factorization goes, launch by launch: the kernel trace of a bench run
(rocprofv3 --kernel-trace CSV, scripts/gpu_trace.sh) joined with the exported program of the same
configuration.  Dispatches are numbered in submission order, and the engine submits one kernel per
launch with work in program order, so the k-th dispatch after k_scatter_val IS the k-th such launch.
Prints, per level, when its first kernel started and its last kernel ended (ms after the scatter),
per stream the busy time inside the level, and the longest stretches in which the chain stream
(the critical path) had no kernel running although the level was not done.  CPU only.

    python scripts/level_timeline.py gpurun_out/trace_x/t_kernel_trace.csv [config] [table]
(table: per-launch alone durations of `bench.py --profile-out`, to print in-program / alone)
"""
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spllt_amd import api, matgen   # noqa: E402

KIND = {0: "potrf", 1: "update", 2: "gather", 3: "exchange", 4: "chain", 5: "marker", 6: "panel", 7: "chainp", 8: "chain2", 9: "trsm2"}


def main():
    trace, config = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nd24k_like")
    table = sys.argv[3] if len(sys.argv) > 3 else None
    A, order, cfg = matgen.build_config(config, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    L = f.program("launches")
    work = [i for i in range(len(L)) if L[i][3] > 0 and L[i][0] != 3]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if "k_scatter_val" in r["Kernel_Name"]]
    a = starts[-1]
    seg = rows[a + 1:a + 1 + len(work)]
    assert len(seg) == len(work), f"trace holds {len(seg)} dispatches after the last scatter, the program {len(work)} launches"
    t0 = int(rows[a]["End_Timestamp"])
    s = np.array([int(r["Start_Timestamp"]) - t0 for r in seg]) / 1e3      # us
    e = np.array([int(r["End_Timestamp"]) - t0 for r in seg]) / 1e3
    P = L[work]
    alone = None
    if table:
        tab = [ln.split() for ln in open(table).read().strip().splitlines()[1:]]
        if len(tab) == len(L):
            alone = np.array([max(float(tab[i][6]) * 1e3 - 3.0, 0.5) for i in work])
    print(f"{config}: {len(work)} kernels, span {e.max() / 1e3:.2f} ms, sum of durations {(e - s).sum() / 1e3:.2f} ms"
          + (f", alone {alone.sum() / 1e3:.2f} ms" if alone is not None else ""))
    names = {0: "chain", 1: "bulk", 2: "far", 3: "wide", 4: "side"}
    prev_end = 0.0
    for k in sorted(set(P[:, 1].tolist())):
        sel = P[:, 1] == k
        line = f"  level {k:2d}: {int(sel.sum()):4d} kernels, first start {s[sel].min() / 1e3:7.3f}, last end {e[sel].max() / 1e3:7.3f} ms (+{(e[sel].max() - prev_end) / 1e3:6.3f})"
        for st in sorted(set(P[sel, 6].tolist())):
            q = sel & (P[:, 6] == st)
            line += f" | {names.get(int(st), st)} {int(q.sum())}: {(e[q] - s[q]).sum() / 1e3:.2f}"
            if alone is not None:
                line += f" ({alone[q].sum() / 1e3:.2f})"
        print(line)
        prev_end = max(prev_end, e[sel].max())
    # the chain stream: time between the end of one of its kernels and the start of the next
    ch = np.where(P[:, 6] == 0)[0]
    gaps = s[ch[1:]] - e[ch[:-1]]
    print(f"  chain stream: {len(ch)} kernels, {(e[ch] - s[ch]).sum() / 1e3:.2f} ms running, {gaps.clip(0).sum() / 1e3:.2f} ms between them"
          + (f" (alone {alone[ch].sum() / 1e3:.2f} ms)" if alone is not None else ""))
    big = np.argsort(-gaps)[:12]
    for g in sorted(big):
        i = ch[g + 1]
        print(f"    {gaps[g]:7.1f} us before launch {work[i]} ({KIND.get(int(P[i][0]), P[i][0])}, level {P[i][1]}, count {P[i][3]}), waits {[int(w) for w in P[i][8:12] if w >= 0]}")
    hist = np.histogram(gaps, bins=[-1e9, 2, 4, 8, 16, 32, 64, 1e9])[0]
    print("    gaps <2 / 2-4 / 4-8 / 8-16 / 16-32 / 32-64 / >64 us:", hist.tolist())
    if alone is not None:
        slow = (e - s) - alone
        for st in (0, 1, 2):
            q = P[:, 6] == st
            print(f"  stream {names[st]}: in program {(e[q] - s[q]).sum() / 1e3:.2f} ms, alone {alone[q].sum() / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
