// Times k_chain_panel alone on one 256 x 256 diagonal sub-tile (panels 0..3) with in-kernel
// phase stamps.  Build (cross-compiles without a GPU), then run on the box:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -DPOTRF_STAMPS -I include -I spllt_amd/csrc scripts/chain_bench.hip -o bin_tmp/chain_bench
#include "../spllt_amd/csrc/kernels.hip"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace spx;
int main() {
  const int w = 256, pw = 64;
  std::vector<double> h((size_t)w * w);
  for (int i = 0; i < w; ++i)
    for (int j = 0; j < w; ++j) h[(size_t)i * w + j] = (i == j) ? w + 1.0 : 1.0 / (1 + abs(i - j));
  double *dA, *dinv; int* flag; ChainUnit* du;
  hipMalloc(&dA, h.size() * 8); hipMalloc(&dinv, h.size() * 8 * 4); hipMalloc(&flag, 4);
  std::vector<ChainUnit> u(4);
  for (int q = 0; q < 4; ++q) {
    u[q].off = 0; u[q].winv_off = winv_offset(w, pw, 256, q); u[q].ld = w; u[q].c0 = q * pw; u[q].pn = pw;
    u[q].cs = 0; u[q].ce = w; u[q].gcol = q * pw;
  }
  hipMalloc(&du, sizeof(ChainUnit) * 4);
  hipMemcpy(du, u.data(), sizeof(ChainUnit) * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[21] = {"start", "load", "J0:A1", "J0:chol16", "J0:subst", "J1:A1", "J1:chol16", "J1:subst",
                           "J2:A1", "J2:chol16", "J2:subst", "J3:A1", "J3:chol16", "J3:subst", "B0", "B recurrences",
                           "store", "sync", "W", "TRSM", "update"};
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int q = 0; q < 4; ++q) {
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch_chain_panel(0, du + q, 1, w - (q + 1) * pw, dA, dinv, flag);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep < 2) continue;
      printf("k_chain_panel q=%d: %.1f us\n", q, ms * 1e3);
#ifdef POTRF_STAMPS
      std::vector<unsigned long long> st(32);
      hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_potrf_stamps), sizeof(unsigned long long) * 32);
      for (int i = 1; i <= 20; ++i) printf("  %-14s %7llu", names[i], st[i] - st[i - 1]), (i % 4 == 0 ? printf("\n") : 0);
      printf("  total %llu cycles (s_memtime, 100 MHz ticks?)\n", st[20] - st[0]);
#endif
    }
  }
  return 0;
}
