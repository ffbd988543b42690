"""per-launch alone times of a configuration WITHOUT factorizing first (for timing-only builds of the
library whose numbers are wrong on purpose, e.g. -DSCATTER_EXPERIMENT=n): prints the sums per
category and, for the inter-node updates, per level"""
import sys, collections, numpy as np
sys.path.insert(0, ".")
from spllt_amd import api, matgen
name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
A, order, cfg = matgen.build_config(name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, order=order)
L = f.program("launches"); units, tiles = f.program("units"), f.program("tiles")
ms = np.minimum(f.profile(val), f.profile(val))
acc = collections.defaultdict(float)
for l, t in zip(L, ms):
    if l[0] == 1 and l[3] > 0:
        u = units[int(tiles[int(l[2])]["unit"])]
        key = ("between", int(l[1])) if u["mode"] == 1 else ("other updates", -1)
    else:
        key = ("not an update", -1)
    acc[key] += t
for k in sorted(acc): print(f"{k[0]:14s} level {k[1]:2d}: {acc[k]:7.3f} ms")
print(f"between, all levels: {sum(v for k, v in acc.items() if k[0] == 'between'):.3f} ms")
