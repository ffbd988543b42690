import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from spllt_amd import matgen
from helpers import make_case
A = matgen.poisson3d(40)
res = {}
for fl in (0, 512):
    f, val = make_case(A, nb=384, nemin=32, engine_flags=fl)
    res[fl] = f.factor(val).wait().get_factor()
    keep = f
d = np.abs(res[0] - res[512])
print("max diff", d.max())
off, bw, bnr, node = keep.sym("bcol_off"), keep.sym("bcol_width"), keep.sym("bcol_nrow"), keep.sym("bcol_node")
lev = keep.sym("level")
bad = np.nonzero(d > 1e-10)[0]
print("nbad", len(bad))
b = np.searchsorted(off, bad, side="right") - 1
import collections
cnt = collections.Counter(b.tolist())
for bb, c in sorted(cnt.items())[:20]:
    idx = bad[b == bb] - off[bb]
    r, cc = idx // bw[bb], idx % bw[bb]
    print("bcol", bb, "node", node[bb], "level", lev[node[bb]], "w", bw[bb], "nrow", bnr[bb], "nbad", c, "rows", r.min(), r.max(), "cols", cc.min(), cc.max(),
          "colset", sorted(set((cc // 64).tolist())), "rowblocks", sorted(set((r // 64).tolist()))[:12])
L = keep.program("launches")
print("kinds", collections.Counter(L[:, 0].tolist()))
