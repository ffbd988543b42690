// Times k_potrf_panel alone (one 64x64 SPD block per workgroup) with in-kernel
// phase stamps.  Build on the box:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -DPOTRF_STAMPS -I include -I spllt_amd/csrc scripts/potrf_bench.hip -o /tmp/potrf_bench
#include "../spllt_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace spx;
int main() {
  const int n = 64, nblk = 64;  // nblk independent blocks
  std::vector<double> h((size_t)nblk * n * n);
  for (int b = 0; b < nblk; ++b)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) h[(size_t)b * n * n + i * n + j] = (i == j) ? n + 1.0 : 1.0 / (1 + abs(i - j));
  double *dA, *dinv; int* flag; PotrfUnit* du; unsigned long long* dst;
  hipMalloc(&dA, h.size() * 8); hipMalloc(&dinv, h.size() * 8); hipMalloc(&flag, 4);
  hipMalloc(&dst, 8 * 16 * nblk);
  std::vector<PotrfUnit> u(nblk);
  for (int b = 0; b < nblk; ++b) { u[b].off = (int64_t)b * n * n; u[b].dinv_off = (int64_t)b * n * n; u[b].ld = n; u[b].n = n; u[b].gcol = 0; u[b].flags = 0; }
  hipMalloc(&du, sizeof(PotrfUnit) * nblk);
  hipMemcpy(du, u.data(), sizeof(PotrfUnit) * nblk, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 64}) {
    float best = 1e9;
    for (int r = 0; r < 10; ++r) {
      hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch_potrf(0, du, grid, dA, dinv, flag);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("k_potrf_panel n=64 grid=%d: %.1f us\n", grid, best * 1e3);
  }
#ifdef POTRF_STAMPS
  std::vector<unsigned long long> st(32);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_potrf_stamps), sizeof(unsigned long long) * 32);
  const char* names[17] = {"start", "load", "J0:A1", "J0:chol16", "J0:subst", "J1:A1", "J1:chol16", "J1:subst",
                           "J2:A1", "J2:chol16", "J2:subst", "J3:A1", "J3:chol16", "J3:subst", "B0 diag inverses",
                           "B recurrences", "store"};
  for (int i = 1; i <= 16; ++i) printf("  %-18s %7llu cycles\n", names[i], st[i] - st[i - 1]);
  printf("  total %llu cycles\n", st[16] - st[0]);
#endif
  return 0;
}
