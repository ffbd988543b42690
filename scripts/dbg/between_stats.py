"""CPU-only: statistics of the inter-node (SCATTER) update units of a configuration, per tree
level -- how dense the source rows / columns of a unit sit in the 64-entry tiles of their
destination block column (the question behind a destination-centric assembly)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from spllt_amd import api, matgen

name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
A, order, cfg = matgen.build_config(name, scale)
n, ptr, row, val = api.csc_lower_1based(A)
f = api.Factorization(n, ptr, row, nb=cfg["nb"], order=order)
si = f.sym_info()
print(si)
units = f.program("units"); tiles = f.program("tiles"); launches = f.program("launches")
relpos = f.program("relpos"); rlist = f.sym("rlist")
bw = f.sym("bcol_width"); bnode = f.sym("bcol_node"); level = f.sym("level"); br0 = f.sym("bcol_r0")
sc = units[units["mode"] == 1]
print("scatter units", len(sc))
# per unit: K, M, N, level of source, level of dest, row tiles touched, col tiles touched
rows = []
for u in sc:
    K = int(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
    M, N = int(u["M"]), int(u["N"])
    rp = relpos[u["relrow_off"]:u["relrow_off"] + M] - u["d_row0"]
    gc = rlist[u["gcol_off"]:u["gcol_off"] + N] - u["d_col0"]
    rt = np.unique(rp // T); ct = np.unique(gc // T)
    # dest tiles (rt x ct) in the lower part only: row tile*T+T-1 + d_row0 >= ... approximate: count all
    rcount = np.bincount(rp // T - rt[0]); ccount = np.bincount(gc // T - ct[0])
    rcount = rcount[rcount > 0]; ccount = ccount[ccount > 0]
    exp_entries = len(rt) * T * len(ct) * T
    ls = level[bnode[u["src_bcol0"]]]; ld = level[bnode[u["dinv_ld"]]]
    rows.append((ls, ld, M, N, K, len(rt), len(ct), M * N, exp_entries))
r = np.array(rows, dtype=np.int64)
print("level nunits  sumMN(M)  expanded(M)  ratio  flops(G)  exp_flops(G)  meanK  meanM meanN")
for l in np.unique(r[:, 0]):
    q = r[r[:, 0] == l]
    mn = q[:, 7].sum(); ex = q[:, 8].sum()
    fl = (2 * q[:, 4] * q[:, 7]).sum() / 1e9; fe = (2 * q[:, 4] * q[:, 8]).sum() / 1e9
    print(f"{l:3d} {len(q):7d} {mn/1e6:9.1f} {ex/1e6:9.1f} {ex/max(mn,1):6.2f} {fl:9.1f} {fe:9.1f} {q[:,4].mean():7.1f} {q[:,2].mean():7.1f} {q[:,3].mean():6.1f}")
print("total flops(G, full rect)", (2 * r[:, 4] * r[:, 7]).sum() / 1e9, "expanded", (2 * r[:, 4] * r[:, 8]).sum() / 1e9)
# by dest level distance
print("by (src level, dst level): units, MN(M)")
for l in np.unique(r[:, 0]):
    q = r[r[:, 0] == l]
    s = " ".join(f"{d}:{(q[q[:,1]==d][:,7].sum())/1e6:.1f}" for d in np.unique(q[:, 1]))
    print(l, s)

# ---- destination-centric pieces: (unit x dest tile) compact products --------------------------
print("\npieces for dest tiles TRxTC (compact products, 16x16 MFMA granules)")
for TR, TC in ((64, 64), (128, 64), (128, 128)):
    tot_use = tot_exec = 0.0; npieces = 0
    per_level = {}
    tile_items = {}
    for u in sc:
        K = int(bw[u["src_bcol0"]:u["src_bcol0"] + u["nseg"]].sum())
        M, N = int(u["M"]), int(u["N"])
        rp = relpos[u["relrow_off"]:u["relrow_off"] + M] - u["d_row0"]
        gc = rlist[u["gcol_off"]:u["gcol_off"] + N] - u["d_col0"]
        rt, rcnt = np.unique(rp // TR, return_counts=True)
        ct, ccnt = np.unique(gc // TC, return_counts=True)
        # piece (a, b): rcnt[a] x ccnt[b]; lower cut: source row index i >= col index j (src_r0 == src_c0)
        rstart = np.concatenate(([0], np.cumsum(rcnt)[:-1])); cstart = np.concatenate(([0], np.cumsum(ccnt)[:-1]))
        ls = int(level[bnode[u["src_bcol0"]]])
        for a in range(len(rt)):
            i0, i1 = rstart[a], rstart[a] + rcnt[a]
            for b in range(len(ct)):
                j0, j1 = cstart[b], cstart[b] + ccnt[b]
                if i1 - 1 < j0: continue
                # useful entries (lower cut)
                jj = np.arange(j0, j1)
                use = np.maximum(0, i1 - np.maximum(i0, jj)).sum()
                gr = -(-(i1 - i0) // 16) * -(-(j1 - j0) // 16) * 256
                tot_use += 2.0 * K * use; tot_exec += 2.0 * K * gr; npieces += 1
                pl = per_level.setdefault(ls, [0.0, 0.0, 0])
                pl[0] += 2.0 * K * use; pl[1] += 2.0 * K * gr; pl[2] += 1
                key = (int(u["dinv_ld"]), int(rt[a]), int(ct[b]), ls)
                tile_items[key] = tile_items.get(key, 0) + 1
    cnt = np.array(list(tile_items.values()))
    print(f"T={TR}x{TC}: pieces {npieces}, useful {tot_use/1e9:.1f} GF, executed {tot_exec/1e9:.1f} GF ({tot_exec/tot_use:.2f}x); dest (tile,level) pairs {len(cnt)}, items/tile mean {cnt.mean():.2f} max {cnt.max()}")
    for l in sorted(per_level):
        p = per_level[l]
        print(f"   level {l}: useful {p[0]/1e9:7.1f} executed {p[1]/1e9:7.1f} ({p[1]/max(p[0],1):.2f}x) pieces {p[2]}")
