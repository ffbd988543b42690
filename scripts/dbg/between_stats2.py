"""CPU-only: executed MFMA work of a destination-centric assembly whose accumulators are in
DESTINATION coordinates, a source contributing to every 16x16 destination fragment it touches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spllt_amd import api, matgen
name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
A, order, cfg = matgen.build_config(name, scale)
n, ptr, row, val = api.csc_lower_1based(A)
f = api.Factorization(n, ptr, row, nb=cfg["nb"], order=order)
units = f.program("units"); relpos = f.program("relpos"); rlist = f.sym("rlist")
bw = f.sym("bcol_width"); bnode = f.sym("bcol_node"); level = f.sym("level")
sc = units[units["mode"] == 1]
G = 16
res = {}
runlens = []
for u in sc:
    K = int(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
    M, N = int(u["M"]), int(u["N"])
    rp = relpos[u["relrow_off"]:u["relrow_off"] + M] - u["d_row0"]
    gc = rlist[u["gcol_off"]:u["gcol_off"] + N] - u["d_col0"]
    rg = np.unique(rp // G); cg = np.unique(gc // G)
    # fragments (a, b) with a-row-group possibly below/at the col group (lower part): count pairs where
    # max dest row of group >= min dest col ... use source order: row index of first row in group vs col index
    # simple: count all pairs whose last source row index >= first source col index
    first_r = np.searchsorted(rp // G, rg, side="left"); last_r = np.searchsorted(rp // G, rg, side="right") - 1
    first_c = np.searchsorted(gc // G, cg, side="left")
    cnt = (last_r[:, None] >= first_c[None, :]).sum()
    use = M * N - N * (N - 1) // 2
    ls = int(level[bnode[u["src_bcol0"]]])
    r = res.setdefault(ls, [0.0, 0.0])
    r[0] += 2.0 * K * use; r[1] += 2.0 * K * cnt * G * G
    d = np.diff(rp); runlens.append((M, 1 + int((d != 1).sum())))
tu = te = 0
for l in sorted(res):
    print(f"level {l}: useful {res[l][0]/1e9:8.1f} GF  executed(dest 16x16 frags) {res[l][1]/1e9:8.1f} GF  {res[l][1]/res[l][0]:.2f}x")
    tu += res[l][0]; te += res[l][1]
print(f"total {tu/1e9:.1f} -> {te/1e9:.1f}  {te/tu:.2f}x")
rl = np.array(runlens)
print("rows per unit / runs per unit: mean run length", rl[:, 0].sum() / rl[:, 1].sum())
# ---- per (src level, dst level) expansion at 16x16 and 64x64 dest granularity, and share of entries
tab = {}
for u in sc:
    K = int(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
    M, N = int(u["M"]), int(u["N"])
    rp = relpos[u["relrow_off"]:u["relrow_off"] + M] - u["d_row0"]
    gc = rlist[u["gcol_off"]:u["gcol_off"] + N] - u["d_col0"]
    ls = int(level[bnode[u["src_bcol0"]]]); ld = int(level[bnode[u["dinv_ld"]]])
    t = tab.setdefault((ls, ld - ls), [0.0, 0.0, 0.0, 0.0])
    use = M * N - N * (N - 1) // 2
    t[0] += use; t[1] += 2.0 * K * use
    for gi, G in enumerate((16, 64)):
        rg = np.unique(rp // G); cg = np.unique(gc // G)
        last_r = np.searchsorted(rp // G, rg, side="right") - 1
        first_c = np.searchsorted(gc // G, cg, side="left")
        cnt = (last_r[:, None] >= first_c[None, :]).sum()
        t[2 + gi] += 2.0 * K * cnt * G * G
print("src level, distance: entries(M) useful GF, x16, x64")
for k in sorted(tab):
    t = tab[k]
    print(f"  {k[0]} +{k[1]}: {t[0]/1e6:7.1f}M {t[1]/1e9:7.1f} GF  {t[2]/t[1]:.2f}x  {t[3]/t[1]:.2f}x")
for dist in range(1, 10):
    e = sum(t[0] for k, t in tab.items() if k[1] == dist); fl = sum(t[1] for k, t in tab.items() if k[1] == dist)
    x16 = sum(t[2] for k, t in tab.items() if k[1] == dist); x64 = sum(t[3] for k, t in tab.items() if k[1] == dist)
    if e: print(f"distance {dist}: entries {e/1e6:.1f}M useful {fl/1e9:.1f} GF x16 {x16/fl:.2f} x64 {x64/fl:.2f}")
