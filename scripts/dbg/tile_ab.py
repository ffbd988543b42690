"""64-tiles everywhere vs the default tile choice on one configuration (resident factorization time)"""
import sys, time, torch
sys.path.insert(0, ".")
from spllt_amd import api, matgen
name = sys.argv[1] if len(sys.argv) > 1 else "serena_like"
A, order, cfg = matgen.build_config(name, 1.0)
n, ptr, row, val = api.csc_lower_1based(A)
dval = torch.tensor(val, device="cuda")
for tile in (None, 64):
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, order=order, tile=tile)
    for _ in range(2):
        f.factor_dev(dval.data_ptr()); f.wait()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f.factor_dev(dval.data_ptr()); f.wait(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, "tile", tile or "default(128 for large launches)", "ms", [round(t, 1) for t in ts], "TFLOP/s", round(f.sym_info()["flops"] / min(ts) / 1e9, 1), flush=True)
    f.close(); del f
    torch.cuda.empty_cache()
