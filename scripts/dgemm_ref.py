"""rocBLAS DGEMM (through torch.matmul, fp64) on the shapes of the update kernel: the tuned
library's rate is the practical ceiling the hand-written k_update has to be read against
(C = A * B^T, A: M x K, B: N x K, row-major = the 'TN' case of a column-major BLAS)."""
import time
import torch

torch.backends.cuda.matmul.allow_tf32 = False
for (M, N) in ((8192, 8192), (2048, 2048)):
    for K in (64, 128, 256, 512, 1024, 4096):
        A = torch.randn(M, K, dtype=torch.float64, device="cuda")
        B = torch.randn(N, K, dtype=torch.float64, device="cuda")
        C = torch.zeros(M, N, dtype=torch.float64, device="cuda")
        # sustained rate, like scripts/update_bench.hip: ~150 ms of back-to-back calls first (the chip
        # needs that long under load to reach its clocks), then the average of a timed train
        est_ms = 2.0 * M * N * K / 40e12 * 1e3 + 0.01
        nwarm, nrep = int(150.0 / est_ms) + 1, int(60.0 / est_ms) + 3
        for _ in range(nwarm):
            torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)
        e1.record()
        torch.cuda.synchronize()
        best = e0.elapsed_time(e1) / nrep
        print(f"M=N={M} K={K:5d}  {best * 1e3:9.1f} us  {2.0 * M * N * K / best / 1e9:7.2f} TFLOP/s", flush=True)
