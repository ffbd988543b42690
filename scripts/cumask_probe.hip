// Does a CU-masked stream leave CUs free for a latency-critical kernel on another stream?
//   hipcc --offload-arch=gfx950 -O2 scripts/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
// A "bulk" kernel (2 workgroups per CU by LDS, ~60 us each, 20 rounds) runs on a stream whose
// CU mask excludes R CUs; while it runs, a one-workgroup "chain" kernel (100 KB LDS, like a
// panel-chain kernel) is launched on an unmasked high-priority stream and its host-visible
// latency (launch -> stream sync) is measured.  Also reports which (xcc, se, cu) the bulk
// kernel's workgroups ran on, i.e. whether and how the mask took effect.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_busy(unsigned* where, int us) {
  extern __shared__ double pad[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    where[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
    pad[0] = 1.0;
  }
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  while (wall_clock64() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
}

__global__ void k_chain(unsigned* where, unsigned long long* stamp) {
  extern __shared__ double pad[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    where[0] = (xcc & 0xf) << 16 | (hw & 0xffff);
    pad[0] = 2.0;
    stamp[0] = wall_clock64();
  }
}

static double now_us() {
  using namespace std::chrono;
  return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, ncu);
  CHK(hipFuncSetAttribute((const void*)k_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int plo, phi;
  CHK(hipDeviceGetStreamPriorityRange(&plo, &phi));
  hipStream_t chain;
  CHK(hipStreamCreateWithPriority(&chain, hipStreamNonBlocking, phi));
  const int nb = ncu * 2 * 20;
  unsigned *d_where, *d_cw;
  unsigned long long* d_stamp;
  CHK(hipMalloc(&d_where, sizeof(unsigned) * nb));
  CHK(hipMalloc(&d_cw, sizeof(unsigned)));
  CHK(hipMalloc(&d_stamp, sizeof(unsigned long long)));
  std::vector<unsigned> where(nb);
  for (int mode = 0; mode < 5; ++mode) {
    // mode 0: no bulk at all; 1: bulk on a plain low-priority stream; 2..4: bulk masked off R CUs
    // with three candidate bit layouts
    hipStream_t bulk = nullptr;
    const int R = 16;
    const char* what = "";
    if (mode <= 1) {
      CHK(hipStreamCreateWithPriority(&bulk, hipStreamNonBlocking, plo));
      what = mode == 0 ? "no bulk kernel" : "bulk unmasked";
    } else {
      std::vector<uint32_t> mask((ncu + 31) / 32, 0xffffffffu);
      if (mode == 2) {          // the first R bits off
        for (int i = 0; i < R; ++i) mask[i / 32] &= ~(1u << (i % 32));
        what = "bulk masked: bits 0..R-1 off";
      } else if (mode == 3) {   // the last R bits off
        for (int i = ncu - R; i < ncu; ++i) mask[i / 32] &= ~(1u << (i % 32));
        what = "bulk masked: bits ncu-R..ncu-1 off";
      } else {                  // every (ncu/R)-th bit off
        for (int i = 0; i < ncu; i += ncu / R) mask[i / 32] &= ~(1u << (i % 32));
        what = "bulk masked: every 16th bit off";
      }
      CHK(hipExtStreamCreateWithCUMask(&bulk, (uint32_t)mask.size(), mask.data()));
    }
    CHK(hipMemset(d_where, 0xff, sizeof(unsigned) * nb));
    CHK(hipDeviceSynchronize());
    if (mode >= 1) hipLaunchKernelGGL(k_busy, dim3(nb), dim3(256), 64 * 1024, bulk, d_where, 60);
    double t_end = now_us() + 150;
    while (now_us() < t_end) {}
    double lat[12];
    for (int i = 0; i < 12; ++i) {
      const double t0 = now_us();
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(256), 100 * 1024, chain, d_cw, d_stamp);
      CHK(hipStreamSynchronize(chain));
      lat[i] = now_us() - t0;
      t_end = now_us() + 30;
      while (now_us() < t_end) {}
    }
    const double t_chain_done = now_us();
    CHK(hipStreamSynchronize(bulk));
    const double t_bulk_done = now_us();
    CHK(hipMemcpy(where.data(), d_where, sizeof(unsigned) * nb, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    std::set<unsigned> xccs;
    for (unsigned v : where)
      if (v != 0xffffffffu) {
        cus.insert(((v >> 16) << 16) | (v & 0xff00));  // xcc | se/sh/cu bits of HW_ID
        xccs.insert(v >> 16);
      }
    printf("mode %d (%s): chain latency us:", mode, what);
    for (int i = 0; i < 12; ++i) printf(" %.0f", lat[i]);
    printf("  | bulk still running after the probes: %s (%.0f us) | bulk ran on %zu distinct CUs in %zu XCCs\n",
           t_bulk_done - t_chain_done > 20 ? "yes" : "NO", t_bulk_done - t_chain_done, cus.size(), xccs.size());
    CHK(hipStreamDestroy(bulk));
  }
  return 0;
}
