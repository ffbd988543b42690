#!/usr/bin/env python3
"""What the stream DAG of the bench workload allows: makespan of the exported program when every
launch takes the time it takes ALONE on the chip (per-launch table of `bench.py --profile-out`,
minus the ~3 us of the two event records around a profiled launch) and the streams never slow each
other down -- list scheduling over the program's own streams, waits and records, 1.4 us per kernel
boundary, 3 us per event record (scripts/gap_probe.hip).  CPU only.

    python scripts/dag_bound.py profiles/r03/launch_table_nd24k_like.txt [config]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spllt_amd import api, matgen   # noqa: E402


def main():
    table, config = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nd24k_like")
    A, order, cfg = matgen.build_config(config, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    L = f.program("launches")
    tab = [ln.split() for ln in open(table).read().strip().splitlines()[1:]]
    assert len(tab) == len(L), "the table belongs to another program (engine flags / knobs?)"
    dur = np.array([max(float(t[6]) * 1e3 - 3.0, 0.5) if int(t[3]) > 0 else 0.0 for t in tab])   # us
    gap, rec_cost = 1.4, 3.0
    rec_t, stream_t, end = {}, {}, np.zeros(len(L))
    for i, l in enumerate(L):
        st = int(l[6])
        ready = stream_t.get(st, 0.0)
        for w in l[8:12]:
            if w >= 0:
                ready = max(ready, rec_t[int(w)])
        end[i] = ready + (dur[i] + gap if dur[i] > 0 else 0.0)
        stream_t[st] = end[i]
        if l[7] >= 0:
            rec_t[int(l[7])] = end[i] + rec_cost
    print(f"{config}: {len(L)} launches; makespan the DAG allows (alone durations, no contention): {end.max() / 1e3:.2f} ms")
    for st, name in ((0, "chain"), (1, "bulk"), (2, "far")):
        sel = L[:, 6] == st
        print(f"  {name:5s} stream: {int((dur[sel] > 0).sum()):4d} launches, {dur[sel].sum() / 1e3:6.2f} ms of kernels alone")
    thr = L[:, 6] != 0
    print(f"  the throughput launches (bulk + far) alone, back to back: {dur[thr].sum() / 1e3:.2f} ms")
    for k in sorted(set(L[:, 1].tolist())):
        if k >= 0:
            print(f"  level {k}: done at {end[L[:, 1] == k].max() / 1e3:6.2f} ms")


if __name__ == "__main__":
    main()
