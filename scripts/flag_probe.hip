// What does it cost to hand data from one workgroup to the others INSIDE a kernel on this chip
// (8 XCDs, one L2 each, coherent only through memory)?  A chain of `steps` hand-offs: at step s the
// producer workgroup (s mod nwg) writes a 32 KB block (a 64 x 64 fp64 tile), releases a flag
// (agent scope); every workgroup acquires the flag, reads the block (checks it), and goes on to
// step s + 1.  Every spin is bounded (the kernel gives up and reports instead of hanging).
// Prints us per hand-off: the price a cooperative panel kernel would pay per panel step instead of
// a kernel boundary (1.4 us) + the launch latency of the next kernel.
//   hipcc --offload-arch=gfx950 -O3 scripts/flag_probe.hip -o bin_tmp/flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// lean variant: the block written with nontemporal (write-through) stores, ONE release store of the
// flag by one thread behind a workgroup barrier, readers poll with relaxed loads and read the block
// with nontemporal loads -- no fence by the readers at all
__global__ __launch_bounds__(256) void k_handoff_lean(double* data, int* flags, int steps, int* gave_up, long long* t_out) {
  const int nwg = gridDim.x, me = blockIdx.x, tid = threadIdx.x;
  long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    double* blk = data + (size_t)(s & 1) * 4096;
    if (me == s % nwg) {
      for (int e = tid; e < 4096; e += 256) __builtin_nontemporal_store((double)(s + 1) + e * 1e-6, &blk[e]);
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flags[s], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(&flags[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (++spins > (1 << 22)) { atomicExch(gave_up, s + 1); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (*(volatile int*)gave_up) return;
    double sum = 0;
    for (int e = tid; e < 4096; e += 256) sum += __builtin_nontemporal_load(&blk[e]) - ((double)(s + 1) + e * 1e-6);
    if (sum != 0.0) atomicExch(gave_up, -(s + 1));      // stale data
  }
  if (tid == 0 && me == 0) *t_out = wall_clock64() - t0;
}

__global__ __launch_bounds__(256) void k_handoff(double* data, int* flags, int steps, int* gave_up, long long* t_out) {
  const int nwg = gridDim.x, me = blockIdx.x, tid = threadIdx.x;
  long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    double* blk = data + (size_t)(s & 1) * 4096;
    if (me == s % nwg) {
      for (int e = tid; e < 4096; e += 256) blk[e] = (double)(s + 1) + e * 1e-6;
      __threadfence();                 // the block is in memory before the flag
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flags[s], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(&flags[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (++spins > (1 << 22)) { atomicExch(gave_up, s + 1); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    __threadfence();                   // (acquire for the whole workgroup)
    if (*(volatile int*)gave_up) return;
    double sum = 0;
    for (int e = tid; e < 4096; e += 256) sum += __builtin_nontemporal_load(&blk[e]) - ((double)(s + 1) + e * 1e-6);
    if (sum != 0.0) atomicExch(gave_up, -(s + 1));      // stale data
  }
  if (tid == 0 && me == 0) *t_out = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 256;
  double* data; int* flags; int* gave_up; long long* t;
  (void)hipMalloc(&data, 2 * 4096 * sizeof(double));
  (void)hipMalloc(&flags, steps * sizeof(int));
  (void)hipMalloc(&gave_up, sizeof(int));
  (void)hipMalloc(&t, sizeof(long long));
  for (int lean = 0; lean < 2; ++lean)
  for (int nwg : {2, 8, 32, 64, 119}) {
    float best = 1e9f; int gu = 0; long long ticks = 0;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipMemset(flags, 0, steps * sizeof(int));
      (void)hipMemset(gave_up, 0, sizeof(int));
      (void)hipMemset(data, 0, 2 * 4096 * sizeof(double));
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (lean) hipLaunchKernelGGL(k_handoff_lean, dim3(nwg), dim3(256), 0, 0, data, flags, steps, gave_up, t);
      else hipLaunchKernelGGL(k_handoff, dim3(nwg), dim3(256), 0, 0, data, flags, steps, gave_up, t);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(&gu, gave_up, sizeof(int), hipMemcpyDeviceToHost);
      (void)hipMemcpy(&ticks, t, sizeof(long long), hipMemcpyDeviceToHost);
      if (ms < best) best = ms;
      if (gu) break;
    }
    printf("%s %3d workgroups, %d hand-offs of 32 KB: %.2f us per hand-off (kernel %.1f us)%s\n", lean ? "lean  " : "fenced", nwg, steps,
           best * 1e3 / steps, best * 1e3, gu > 0 ? "  GAVE UP (flag never seen)" : gu < 0 ? "  STALE DATA" : "");
  }
  return 0;
}
