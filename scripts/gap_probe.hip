// Where do the ~6 us between dependent launches on the chain stream come from?
// (hipcc --offload-arch=gfx950 -O3 scripts/gap_probe.hip -o bin_tmp/gap_probe)
// Period of a chain of dependent one-workgroup launches, by stream kind, by what sits between
// two launches (nothing / an event record nobody waits for / a wait for an already fired event),
// with kernels that differ in their LDS footprint, alone on the chip and beside a long-running
// kernel on a second (CU-masked) stream, and as a hipGraph of kernel nodes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// (each probe kernel spins ~8 us: with shorter kernels the HOST's launch rate is what is measured
// -- 2.7 us per launch, 6.4 with an event record -- not the device's cost of a boundary)
__device__ __forceinline__ void spin(int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_small(double* p) {
  spin(800);
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0;
}
__global__ void k_lds(double* p) {
  extern __shared__ double sm[];
  sm[threadIdx.x] = p[threadIdx.x];
  spin(800);
  __syncthreads();
  if (threadIdx.x == 0) p[blockIdx.x] += sm[1];
}
__global__ void k_static_lds(double* p) {
  __shared__ double sm[8192];
  sm[threadIdx.x] = p[threadIdx.x];
  spin(800);
  __syncthreads();
  if (threadIdx.x == 0) p[blockIdx.x] += sm[1];
}
// keeps `n` workgroups busy for roughly `iters` dependent fp64 FMAs with streaming loads and stores
__global__ void k_busy(double* p, int64_t stride, int iters) {
  double a = p[blockIdx.x * stride + threadIdx.x];
  for (int i = 0; i < iters; ++i) {
    a = __builtin_fma(a, 1.0000001, 1e-9);
    if ((i & 255) == 0) p[blockIdx.x * stride + threadIdx.x + (i & 1023)] = a;
  }
  p[blockIdx.x * stride + threadIdx.x] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  double* d;
  CK(hipMalloc(&d, 1 << 28));
  CK(hipMemset(d, 0, 1 << 28));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s_plain, s_prio, s_mask, s_mask2;
  CK(hipStreamCreateWithFlags(&s_plain, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&s_prio, hipStreamNonBlocking, hi));
  std::vector<uint32_t> mask(8, 0u);
  for (int cu = 0; cu < 224; ++cu) mask[cu / 32] |= 1u << (cu % 32);
  CK(hipExtStreamCreateWithCUMask(&s_mask, 8, mask.data()));
  CK(hipExtStreamCreateWithCUMask(&s_mask2, 8, mask.data()));
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  std::vector<hipEvent_t> evs(2048);
  for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int N = 1000;
  auto period = [&](const char* what, hipStream_t st, int mode, bool busy) -> int {
    // mode 0: same kernel; 1: alternate small / static-LDS / dynamic-LDS kernels; 2: + event record
    // behind every launch; 3: + wait for an event recorded on ANOTHER idle stream long ago;
    // 4: record + wait on the own previous event (what a DAG edge into the same stream costs)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      if (busy) {
        hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s_mask, d + (1 << 20), (int64_t)2048, 12000000);
        hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s_mask2, d + (1 << 23), (int64_t)2048, 12000000);
      }
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < N; ++i) {
        if (mode == 3) CK(hipStreamWaitEvent(st, evs[1500], 0));
        if (mode == 4 && i > 0) CK(hipStreamWaitEvent(st, evs[i - 1], 0));
        const int k = mode == 0 ? 0 : i % 3;
        if (k == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, st, d);
        else if (k == 1) hipLaunchKernelGGL(k_static_lds, dim3(1), dim3(256), 0, st, d);
        else hipLaunchKernelGGL(k_lds, dim3(1), dim3(256), 40 * 1024, st, d);
        if (mode == 2 || mode == 4) CK(hipEventRecord(evs[i], st));
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("%-58s %s: %6.2f us per launch (of which ~8.0 the kernel)\n", what, busy ? "beside busy streams" : "alone              ", ms * 1e3 / N);
    }
    return 0;
  };
  CK(hipEventRecord(evs[1500], s_plain));
  CK(hipDeviceSynchronize());
  for (int busy = 0; busy < 2; ++busy) {
    period("null stream, one kernel", 0, 0, busy);
    period("non-blocking stream, one kernel", s_plain, 0, busy);
    period("priority stream, one kernel", s_prio, 0, busy);
    period("CU-masked stream, one kernel", s_mask, 0, false);
    period("priority stream, three kernels (LDS footprints differ)", s_prio, 1, busy);
    period("priority stream, three kernels + event record each", s_prio, 2, busy);
    period("priority stream, three kernels + wait on a fired event", s_prio, 3, busy);
    period("priority stream, record + wait on own previous event", s_prio, 4, busy);
  }
  // the same chain as a graph of kernel nodes with explicit dependencies
  {
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t prev = nullptr;
    void* args[] = {&d};
    for (int i = 0; i < N; ++i) {
      hipKernelNodeParams kp{};
      const int k = i % 3;
      kp.func = k == 0 ? (void*)k_small : k == 1 ? (void*)k_static_lds : (void*)k_lds;
      kp.gridDim = dim3(1);
      kp.blockDim = dim3(256);
      kp.sharedMemBytes = k == 2 ? 40 * 1024 : 0;
      kp.kernelParams = args;
      hipGraphNode_t n;
      CK(hipGraphAddKernelNode(&n, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
      prev = n;
    }
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int busy = 0; busy < 2; ++busy)
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        if (busy) {
          hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s_mask, d + (1 << 20), (int64_t)2048, 12000000);
          hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s_mask2, d + (1 << 23), (int64_t)2048, 12000000);
        }
        CK(hipEventRecord(e0, s_prio));
        CK(hipGraphLaunch(ge, s_prio));
        CK(hipEventRecord(e1, s_prio));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%-58s %s: %6.2f us per node\n", "graph of 1000 dependent kernel nodes (priority stream)", busy ? "beside busy streams" : "alone              ", ms * 1e3 / N);
      }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
