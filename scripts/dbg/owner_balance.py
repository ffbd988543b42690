"""CPU-only: balance of the subtree partition (assign_owners) under the time model and under flops,
for both weightings (SPLLT_OWNER_MODEL=flops|time)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 3:
    import numpy as np
    from spllt_amd import api, matgen
    name, scale, w = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    A, order, cfg = matgen.build_config(name, scale)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], order=order, prune_tree=True, ncpu=w)
    f.set_partition(0, w)
    owner = f.partition("owner")
    sptr, rptr = f.sym("sptr"), f.sym("rptr")
    nn = len(owner)
    ncol = np.diff(sptr)[:nn].astype(np.float64); nrow = np.diff(rptr)[:nn].astype(np.float64)
    fl = np.array([sum((m - k + j) ** 2 for j in range(1, int(k) + 1)) for m, k in zip(nrow, ncol)])
    t = fl / (68e12 * ncol / (ncol + 90.0)) + 0.5 * (nrow - ncol) ** 2 / 262e9
    out = {"w": w, "top_nodes": int((owner < 0).sum()), "top_flops_frac": float(fl[owner < 0].sum() / fl.sum())}
    out["flops"] = [float(fl[owner == r].sum() / 1e9) for r in range(w)]
    out["time_ms"] = [float(t[owner == r].sum() * 1e3) for r in range(w)]
    out["balance_time"] = min(out["time_ms"]) / max(out["time_ms"])
    out["balance_flops"] = min(out["flops"]) / max(out["flops"])
    print(json.dumps(out))
else:
    name = sys.argv[1] if len(sys.argv) > 1 else "flan_like"
    scale = sys.argv[2] if len(sys.argv) > 2 else "1.0"
    for model in ("flops", "time"):
        for w in (2, 4, 8):
            r = subprocess.run([sys.executable, __file__, name, scale, str(w)], capture_output=True, text=True,
                               env=dict(os.environ, SPLLT_OWNER_MODEL=model))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
            print(model, json.dumps({k: (round(v, 3) if isinstance(v, float) else ([round(x, 1) for x in v] if isinstance(v, list) else v)) for k, v in d.items()}))
