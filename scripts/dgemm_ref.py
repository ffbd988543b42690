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
        for _ in range(3):
            torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        print(f"M=N={M} K={K:5d}  {best * 1e3:9.1f} us  {2.0 * M * N * K / best / 1e9:7.2f} TFLOP/s", flush=True)
