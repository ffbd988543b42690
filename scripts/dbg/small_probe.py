"""eager vs HIP-graph replay over problem sizes (GPU): where does the replay stop paying?"""
import sys, json
sys.path.insert(0, '.')
import torch
torch.cuda.set_device(0)
import bench
from spllt_amd import matgen
cases = [("p2d48", matgen.poisson2d(48), 32, 16), ("p2d128", matgen.poisson2d(128), 256, 32), ("p3d20", matgen.poisson3d(20), 64, 16),
         ("p3d32", matgen.poisson3d(32), 128, 32), ("nd16", matgen.nd_like((16, 16, 16), 2), 128, 32),
         ("nd24", matgen.nd_like((24, 24, 24), 2), 256, 32), ("p3d48", matgen.poisson3d(48), 256, 32),
         ("nd30", matgen.nd_like((30, 30, 30), 3), 256, 32)]
for label, A, nb, nemin in cases:
    d = bench.run_small_config(label, A, nb, nemin, steps=15)
    print(label, "launches", d["launches"], "GF %.2f" % (d["flops_sym"] / 1e9), {k: d[k]["wall_ms"] for k in ("eager", "graph_chain", "graph_dag")}, flush=True)
