// Launch-latency floor of dependent kernels on one stream (hipcc --offload-arch=gfx950 -O3).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
__global__ void k_touch(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
int main() {
  double* d; hipMalloc(&d, 1 << 24); hipMemset(d, 0, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 128, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, (int*)nullptr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("empty kernel, grid %4d: %.2f us per dependent launch\n", grid, ms);
    }
  }
  for (int grid : {1, 128, 512}) {
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, 0, d, grid * 256);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("load+store kernel, grid %4d: %.2f us per dependent launch\n", grid, ms);
  }
  return 0;
}
