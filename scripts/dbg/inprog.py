"""serial vs in-program time per launch category (and per stream) of one configuration"""
import sys, os, collections, numpy as np, time, torch
sys.path.insert(0, ".")
from spllt_amd import api, matgen
name = sys.argv[1] if len(sys.argv) > 1 else "nd24k_like"
A, order, cfg = matgen.build_config(name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
flags = int(os.environ.get("SPLLT_ENGINE_FLAGS", "0"))
f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, order=order, engine_flags=flags)
dval = torch.tensor(val, device="cuda")
for _ in range(3):
    f.factor_dev(dval.data_ptr()); f.wait()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f.factor_dev(dval.data_ptr()); f.wait(); ts.append((time.perf_counter() - t0) * 1e3)
print("factor ms", np.round(ts, 2))
L = f.program("launches"); units, tiles = f.program("units"), f.program("tiles"); bc_off = f.sym("bcol_off")
def cat(l):
    if l[0] != 1: return {0: "potrf", 4: "chain", 5: "winv", 7: "panel", 6: "gather"}.get(int(l[0]), "other")
    if l[3] == 0: return "marker"
    u = units[int(tiles[int(l[2])]["unit"])]
    if u["mode"] == 2: return "trsm"
    if u["mode"] == 1: return "between"
    if bc_off[int(u["src_bcol0"])] == u["d_off"]: return "inpanel"
    return {0: "next", 1: "trailing"}.get(int(l[6]), "update")
ser = f.profile(val); inp = f.profile(val, in_program=True)
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
st = collections.defaultdict(lambda: [0, 0.0, 0.0])
for l, a, b in zip(L, ser, inp):
    c = cat(l); acc[c][0] += 1; acc[c][1] += a; acc[c][2] += b
    s = int(l[6]); st[s][0] += 1; st[s][1] += a; st[s][2] += b
print("category count serial_ms inprogram_ms")
for c, v in sorted(acc.items()): print(f"{c:10s} {v[0]:5d} {v[1]:8.3f} {v[2]:8.3f}")
print("stream count serial_ms inprogram_ms")
for c, v in sorted(st.items()): print(f"{c:10d} {v[0]:5d} {v[1]:8.3f} {v[2]:8.3f}")
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        for i, (l, a, b) in enumerate(zip(L, ser, inp)):
            fh.write(f"{i} {l[0]} lev {l[1]} cnt {l[3]} tile {l[4]} st {l[6]} {cat(l)} ser {a:.4f} inp {b:.4f}\n")
