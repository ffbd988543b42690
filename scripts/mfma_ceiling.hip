// Practical ceiling of v_mfma_f64_16x16x4_f64 on the whole chip: every wave runs a loop of
// independent MFMAs on register operands (8 accumulators, no memory traffic), W waves per SIMD.
// What fraction of the vendor peak (78.6 TFLOP/s at the boost clock) the matrix pipes sustain
// under an fp64 load is the number the update kernel's asymptote has to be read against.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o bin_tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

int main() {
  double* out;
  hipMalloc(&out, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg_per_cu : {1, 2, 4, 8}) {
    const int grid = 256 * wg_per_cu;   // 256 CUs, 4 waves (one per SIMD) per workgroup
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * iters * 8 * 2048.0;
    printf("waves/SIMD %d: %.2f ms  %.2f TFLOP/s (long run: %d x 8 MFMAs per wave)\n", wg_per_cu, best,
           flops / best / 1e9, iters);
  }
  // a long launch (~1 s) to see the sustained clock
  for (int r = 0; r < 2; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma, dim3(256 * 4), dim3(256), 0, 0, out, iters * 20, 1.0, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("sustained (%.0f ms): %.2f TFLOP/s\n", ms, (double)256 * 4 * 4 * iters * 20 * 8 * 2048.0 / ms / 1e9);
  }
  return 0;
}
