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
  hipMalloc(&dA, h.size() * 8); hipMalloc(&dinv, h.size() * 8); hipMemset(dinv, 0, h.size() * 8); hipMalloc(&flag, 4);
  hipMalloc(&dst, 8 * 16 * nblk);
  std::vector<PotrfUnit> u(nblk);
  for (int b = 0; b < nblk; ++b) { u[b].off = (int64_t)b * n * n; u[b].dinv_off = (int64_t)b * n * n; u[b].ld = n; u[b].n = n; u[b].gcol = 0; u[b].flags = 8; }
  hipMalloc(&du, sizeof(PotrfUnit) * nblk);
  hipMemcpy(du, u.data(), sizeof(PotrfUnit) * nblk, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 64}) {
    float best = 1e9;
    for (int r = 0; r < 10; ++r) {
      hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch_potrf(0, du, grid, dA, dinv, flag, u[0]);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("k_potrf_panel n=64 grid=%d: %.1f us\n", grid, best * 1e3);
  }
  {
    // check of block 0 against a host Cholesky and its inverse
    std::vector<double> L(n * n), W(n * n), ref(h.begin(), h.begin() + n * n);
    hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(W.data(), dinv, n * n * 8, hipMemcpyDeviceToHost);
    for (int j = 0; j < n; ++j) {
      for (int k = 0; k < j; ++k) ref[j * n + j] -= ref[j * n + k] * ref[j * n + k];
      ref[j * n + j] = sqrt(ref[j * n + j]);
      for (int i = j + 1; i < n; ++i) {
        for (int k = 0; k < j; ++k) ref[i * n + j] -= ref[i * n + k] * ref[j * n + k];
        ref[i * n + j] /= ref[j * n + j];
      }
    }
    double el = 0, ew = 0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {
        el = fmax(el, fabs(L[i * n + j] - ref[i * n + j]));
        double sacc = 0;   // (L W)[i][j]
        for (int k = j; k <= i; ++k) sacc += ref[i * n + k] * W[k * n + j];
        ew = fmax(ew, fabs(sacc - (i == j ? 1.0 : 0.0)));
      }
    double eu = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) eu = fmax(eu, fabs(W[i * n + j]));
    printf("check: max|L - L_host| = %.2e, max|L W - I| = %.2e, max|upper(W)| = %.2e\n", el, ew, eu);
  }
#ifdef POTRF_STAMPS
  std::vector<unsigned long long> st(32);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_potrf_stamps), sizeof(unsigned long long) * 32);
#ifdef POTRF_V1
  const char* names[17] = {"start", "load", "J0:A1", "J0:chol16", "J0:subst", "J1:A1", "J1:chol16", "J1:subst",
                           "J2:A1", "J2:chol16", "J2:subst", "J3:A1", "J3:chol16", "J3:subst", "B0 diag inverses",
                           "B recurrences", "store"};
  const int last = 16;
#else
  // round-3 body: wave 0's critical path, everything else in its shadow
  const char* names[12] = {"start", "load", "J0:chol16", "J0:solve below", "J1:upd+chol16", "J1:solve below",
                           "J2:upd+chol16", "J2:solve below", "J3:upd+chol16", "J3:(barriers)", "last inverse row + stores",
                           "last stores issued"};
  const int last = 11;
  printf("  wave 0, J = 0: read block %llu, first pivot %llu, pivot loop %llu, square roots + scaling %llu, write-back %llu (last J)\n",
         st[21] - st[20], 0ull, st[22] - st[21], st[23] - st[22], st[25] - st[24]);
#endif
  for (int i = 1; i <= last; ++i) printf("  %-26s %7llu cycles\n", names[i], st[i] - st[i - 1]);
  printf("  total %llu cycles\n", st[last] - st[0]);
#endif
  return 0;
}
