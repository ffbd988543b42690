"""CPU-only: how many conflict-free colours do the inter-node update launches need?  Two units of
a launch conflict when they write into the same 64 x 64 destination tile of the same block column;
greedy colouring in unit order per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spllt_amd import api, matgen
name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
A, order, cfg = matgen.build_config(name, scale)
n, ptr, row, val = api.csc_lower_1based(A)
f = api.Factorization(n, ptr, row, nb=cfg["nb"], order=order)
units = f.program("units"); tiles = f.program("tiles"); L = f.program("launches")
relpos = f.program("relpos"); rlist = f.sym("rlist")
tot_l = tot_c = 0
for li, l in enumerate(L):
    if l[0] != 1 or l[3] == 0: continue
    tl = tiles[int(l[2]):int(l[2] + l[3])]
    uids = sorted(set(tl["unit"].tolist()))
    if units[uids[0]]["mode"] != 1: continue
    foot = {}
    for uid in uids:
        u = units[uid]
        M, N = int(u["M"]), int(u["N"])
        rp = (relpos[u["relrow_off"]:u["relrow_off"] + M] - u["d_row0"]) // 64
        gc = (rlist[u["gcol_off"]:u["gcol_off"] + N] - u["d_col0"]) // 64
        rt = np.unique(rp); ct = np.unique(gc)
        foot[uid] = set((int(u["dinv_ld"]), int(a), int(b)) for a in rt for b in ct)
    colors = []   # list of sets of tiles used
    ncol_tiles = []
    cnt = {}
    for uid in uids:
        for ci, used in enumerate(colors):
            if not (used & foot[uid]):
                used |= foot[uid]; cnt[uid] = ci; break
        else:
            colors.append(set(foot[uid])); cnt[uid] = len(colors) - 1
    per = [0] * len(colors)
    for t in tl: per[cnt[int(t["unit"])]] += 1
    tot_l += 1; tot_c += len(colors)
    print(f"launch {li} level {l[1]} tiles {l[3]} T{l[4]} units {len(uids)}: colours {len(colors)} tiles/colour {per}")
print("scatter launches", tot_l, "-> coloured launches", tot_c)
