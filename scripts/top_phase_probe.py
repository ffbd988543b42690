import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from spllt_amd import api, matgen
A, order, cfg = matgen.build_config("nd24k_like", 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
for w in (2, 8):
    f = api.Factorization(n, ptr, row, nb=256, nemin=32, prune_tree=True, ncpu=w, order=order)
    xel = f.set_partition(0, w)
    xb = torch.zeros(max(xel, 1), dtype=torch.float64, device="cuda")
    f.set_exchange_buffer(xb.data_ptr())
    ms = np.minimum(f.profile(val), f.profile(val))
    L = f.program("launches")
    ix = int(np.where(L[:, 0] == 2)[0][0])
    owner = f.partition("owner"); lvl = f.sym("level")
    print("w", w, "top nodes", int((owner < 0).sum()), "levels of top nodes", np.unique(lvl[owner < 0], return_counts=True))
    print("  phase1 launches", ix, "ms %.2f" % ms[:ix].sum(), " phase2 launches", len(L) - ix - 1, "ms %.2f" % ms[ix + 1:].sum())
    for lev in np.unique(L[ix + 1:, 1]):
        sel = (np.arange(len(L)) > ix) & (L[:, 1] == lev)
        print("   top level", lev, "launches", int(sel.sum()), "ms %.2f" % ms[sel].sum(), "gflop %.1f" % (L[sel, 5].sum() / 1e9))
    f.close()
