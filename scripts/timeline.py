#!/usr/bin/env python3
"""The real multi-stream program on the clock (GPU): spllt_hip_timeline gives the completion
time of every event the program records itself -- nothing is added to the streams, unlike a
profiler trace (rocprofv3 adds 8-10 us between dependent dispatches) or the bracketing events of
`spllt_hip_profile_in_program` (6 us per launch).  Printed: per level, when its last recorded event
completed; per waiting launch of the chain stream, how long after the previous chain record its own
wait was satisfied (= how long the chain stood still for another stream).

    python scripts/timeline.py [config] [repeats] [level: list every event of it]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spllt_amd import api, matgen   # noqa: E402

KIND = {0: "potrf", 1: "update", 2: "exchange", 4: "chain", 6: "gather", 7: "panel", 8: "chain2", 9: "trsm2"}
NAMES = {0: "chain", 1: "bulk", 2: "far", 3: "side", 4: "wide"}


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    A, order, cfg = matgen.build_config(config, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    f.factor(val).wait()
    L = f.program("launches")
    runs = np.array([f.timeline(val) for _ in range(reps)])
    t = runs[np.argmin(runs[:, -1])]
    nl = len(L)
    print(f"{config}: {nl} launches, program ends at {t[-1]:.3f} ms (runs: {' '.join('%.2f' % r[-1] for r in runs)})")
    rec = {int(L[i][7]): t[i] for i in range(nl) if L[i][7] >= 0}
    prev = 0.0
    for k in sorted(set(L[:, 1].tolist())):
        if k < 0:
            continue
        sel = [i for i in range(nl) if L[i][1] == k and t[i] >= 0]
        if not sel:
            continue
        line = f"  level {k:2d}: last event at {max(t[i] for i in sel):7.3f} ms (+{max(t[i] for i in sel) - prev:6.3f})"
        for st in (0, 1, 2):
            q = [i for i in sel if L[i][6] == st]
            if q:
                line += f" | {NAMES[st]} first {min(t[i] for i in q):7.3f} last {max(t[i] for i in q):7.3f} ({len(q)} events)"
        print(line)
        prev = max(prev, max(t[i] for i in sel))
    # chain-stream launches that wait for an event of ANOTHER stream: when was that event there,
    # when was the chain's previous record there
    last_chain = 0.0
    stalls = []
    for i in range(nl):
        st = int(L[i][6])
        if st in (0, 3, 4):
            ws = [int(w) for w in L[i][8:12] if w >= 0]
            if ws:
                ready = max(rec[w] for w in ws)
                if ready > last_chain + 0.002:
                    stalls.append((ready - last_chain, i, ready))
            if t[i] >= 0:
                last_chain = t[i]
    tot = sum(s[0] for s in stalls)
    print(f"  chain stream stood still for events of other streams (lower bound: since its last own record): {tot:.2f} ms in {len(stalls)} waits")
    for d, i, ready in sorted(stalls, reverse=True)[:16]:
        print(f"    {d * 1e3:7.1f} us before launch {i} ({KIND.get(int(L[i][0]), L[i][0])}, level {L[i][1]}, count {L[i][3]}), ready at {ready:.3f} ms")
    if len(sys.argv) > 3:
        # every event of one level: launch, stream, kind, count, time, the events it waited for
        lev = int(sys.argv[3])
        print(f"  events of level {lev}:")
        for i in range(nl):
            if L[i][1] == lev and t[i] >= 0:
                ws = [f"{rec[int(w)]:.3f}" for w in L[i][8:12] if w >= 0]
                print(f"    launch {i:4d} {NAMES[int(L[i][6])]:5s} {KIND.get(int(L[i][0]), L[i][0]):7s} count {int(L[i][3]):6d} tile {int(L[i][4]):3d}  done {t[i]:8.3f} ms  waited for {ws}")
    f.close()


if __name__ == "__main__":
    main()
