#!/usr/bin/env python3
"""Kernel-trace summary of one bench run (rocprofv3 --kernel-trace CSV): per kernel name the
number of dispatches and the summed duration, the span from the first to the last dispatch of the
LAST factorization (delimited by k_scatter_val) and the idle time inside it per queue."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'k_scatter_val' in r['Kernel_Name']]
if not starts:
    print('no factorization found'); sys.exit(0)
# last complete factorization: from its scatter to the next non-spx kernel or the end
a = starts[-1]
b = a
while b + 1 < len(rows) and ('spx::' in rows[b + 1]['Kernel_Name']) and 'k_solve' not in rows[b + 1]['Kernel_Name'] and 'k_scatter_val' not in rows[b + 1]['Kernel_Name']:
    b += 1
seg = rows[a:b + 1]
t0, t1 = int(seg[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in seg)
print(f'last factorization: {len(seg)} dispatches, span {(t1 - t0) / 1e6:.3f} ms')
by = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r['Kernel_Name'].replace('void ', '').replace('spx::', '')[:40]
    by[n][0] += 1
    by[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = 0
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f'  {n:42s} {c:5d} dispatches {d / 1e6:8.3f} ms')
    tot += d
print(f'  sum of kernel durations {tot / 1e6:.3f} ms')
# union busy time (any kernel running)
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f'  some kernel running {busy / 1e6:.3f} ms, nothing running {(t1 - t0 - busy) / 1e6:.3f} ms')
q = collections.Counter(r['Queue_Id'] for r in seg)
print('  dispatches per queue', dict(q))
