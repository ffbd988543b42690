// Device ceilings used to price the kernels (fp64 MFMA, fp64 VALU FMA, HBM copy).
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench.hip -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// in-kernel clock: shader cycles (s_memtime) per 100 MHz reference tick (s_memrealtime)
template <int NACC, bool MFMA>
__global__ __launch_bounds__(256) void k_clock(double* out, unsigned long long* stamps, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MFMA) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else { acc[i][0] = __builtin_fma(acc[i][0], a, b); acc[i][1] = __builtin_fma(acc[i][1], a, b);
             acc[i][2] = __builtin_fma(acc[i][2], a, b); acc[i][3] = __builtin_fma(acc[i][3], a, b); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) out[i] = in[i];
}

template <class F>
float timeit(F f, int rep = 5) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f();
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < rep; ++r) {
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* out;
  (void)hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 20000;
  for (int blocksPerCU : {1, 2}) {
    int grid = p.multiProcessorCount * blocksPerCU;
    float ms = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    double fl = (double)grid * 4 * iters * 8 * 2048.0;
    printf("mfma_f64_16x16x4 NACC=8 blocks/CU=%d: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at %.2f GHz)\n",
           blocksPerCU, ms, fl / ms / 1e9, ms * 1e-3 * p.clockRate * 1e3 / (iters * 8.0 * blocksPerCU),
           p.clockRate / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_mfma<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    fl = (double)grid * 4 * iters * 2 * 2048.0;
    printf("mfma_f64_16x16x4 NACC=2 blocks/CU=%d: %.3f ms  %.2f TFLOP/s\n", blocksPerCU, ms, fl / ms / 1e9);
  }
  {
    unsigned long long* st;
    (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * 4096);
    std::vector<unsigned long long> h(2 * 4096);
    for (int mf = 1; mf >= 0; --mf)
      for (int blocksPerCU : {1, 2, 4}) {
        int grid = p.multiProcessorCount * blocksPerCU;
        const int it2 = 200000;  // long enough (>= 0.1 s) for the clock to settle
        float ms = timeit([&] {
          if (mf) hipLaunchKernelGGL((k_clock<8, true>), dim3(grid), dim3(256), 0, 0, out, st, it2, 1.0, 1.0);
          else hipLaunchKernelGGL((k_clock<8, false>), dim3(grid), dim3(256), 0, 0, out, st, it2, 1.0000001, 1e-9);
        }, 3);
        (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
        double clk = 0;
        for (int g = 0; g < grid; ++g) clk += (double)h[2 * g] / (double)h[2 * g + 1] * 100e6;
        clk /= grid;
        double fl = mf ? (double)grid * 4 * it2 * 8 * 2048.0 : (double)grid * 256 * it2 * 32 * 2.0;
        double cyc = (double)h[0] / ((double)it2 * 8 * (mf ? 1 : 4));
        printf("%s long run blocks/CU=%d: %.2f ms %.2f TFLOP/s  in-kernel clock %.3f GHz  %.1f shader cycles per wave-instr\n",
               mf ? "mfma_f64" : "v_fma_f64", blocksPerCU, ms, fl / ms / 1e9, clk / 1e9, cyc);
      }
  }
  for (int blocksPerCU : {1, 2, 4}) {
    int grid = p.multiProcessorCount * blocksPerCU;
    float ms = timeit([&] { hipLaunchKernelGGL(k_fma<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    double fl = (double)grid * 256 * iters * 16 * 2.0;
    printf("v_fma_f64 NACC=16 blocks/CU=%d: %.3f ms  %.2f TFLOP/s\n", blocksPerCU, ms, fl / ms / 1e9);
  }
  size_t n = (size_t)1 << 28;  // 2^28 double2 = 4 GiB
  double2 *a, *b;
  (void)hipMalloc(&a, n * 16);
  (void)hipMalloc(&b, n * 16);
  (void)hipMemset(a, 1, n * 16);
  float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, n); });
  printf("copy 4 GiB: %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e9);
  ms = timeit([&] { (void)hipMemsetAsync(b, 0, n * 16, 0); });
  printf("memset 4 GiB: %.3f ms  %.2f TB/s\n", ms, 1.0 * n * 16 / ms / 1e9);
  return 0;
}
