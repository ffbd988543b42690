// k_chain_block + k_trsm_rows alone, with in-kernel stamps (thread 0 of workgroup 0) and a host check.
// Build on the box:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -DCHAIN_STAMPS -I include -I spllt_amd/csrc scripts/chain_block_bench.hip -o /tmp/chain_block_bench
#include "../spllt_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace spx;
int main(int argc, char** argv) {
  const int cw = argc > 1 ? atoi(argv[1]) : 256, M = argc > 2 ? atoi(argv[2]) : 7424, pw = 64;
  const int ld = cw, nrow = cw + M;
  std::vector<double> h((size_t)nrow * ld);
  for (int i = 0; i < nrow; ++i)
    for (int j = 0; j < cw; ++j) h[(size_t)i * ld + j] = (i == j) ? cw + 1.0 : (j > i && i < cw ? 0.0 : 1.0 / (1 + abs(i - j) % 97));
  double *dA, *dinv; int* flag;
  hipMalloc(&dA, h.size() * 8 + 4096); hipMalloc(&dinv, (size_t)cw * cw * 8 + 4096); hipMemset(dinv, 0, (size_t)cw * cw * 8);
  hipMalloc(&flag, 4);
  ChainUnit cu{}; cu.off = 0; cu.winv_off = 0; cu.ld = ld; cu.c0 = 0; cu.pn = cw; cu.cs = 0; cu.ce = cw; cu.gcol = 0;
  ChainUnit* dcu; hipMalloc(&dcu, sizeof cu); hipMemcpy(dcu, &cu, sizeof cu, hipMemcpyHostToDevice);
  UpdUnit u{}; u.d_off = 0; u.d_ld = ld; u.d_row0 = cw; u.d_col0 = 0; u.M = M; u.N = cw; u.dinv_off = 0; u.mode = MODE_TRSM;
  UpdUnit* du; hipMalloc(&du, sizeof u); hipMemcpy(du, &u, sizeof u, hipMemcpyHostToDevice);
  const int nt = (M + kTrsmRows - 1) / kTrsmRows;
  std::vector<UpdTile> tl(nt);
  for (int t = 0; t < nt; ++t) tl[t] = UpdTile{0, (short)t, 0};
  UpdTile* dt; hipMalloc(&dt, sizeof(UpdTile) * nt); hipMemcpy(dt, tl.data(), sizeof(UpdTile) * nt, hipMemcpyHostToDevice);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  LaunchSink sink{}; sink.stream = 0;
  float b1 = 1e9, b2 = 1e9;
  for (int r = 0; r < 10; ++r) {
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dinv, 0, (size_t)cw * cw * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch_chain_block(sink, dcu, 1, dA, dinv, flag, pw, cu);
    hipEventRecord(e1);
    launch_trsm_rows(sink, dt, nt, du, dA, dinv, pw, 1);
    hipEventRecord(e2); hipEventSynchronize(e2);
    float m1, m2; hipEventElapsedTime(&m1, e0, e1); hipEventElapsedTime(&m2, e1, e2);
    b1 = fminf(b1, m1); b2 = fminf(b2, m2);
  }
  printf("cw=%d M=%d: k_chain_block %.1f us, k_trsm_rows (%d workgroups) %.1f us\n", cw, M, b1 * 1e3, nt, b2 * 1e3);
  // host check: Cholesky of the diagonal block, rows below solved
  std::vector<double> L(h.size()), ref(h);
  hipMemcpy(L.data(), dA, h.size() * 8, hipMemcpyDeviceToHost);
  for (int j = 0; j < cw; ++j) {
    for (int k = 0; k < j; ++k) ref[(size_t)j * ld + j] -= ref[(size_t)j * ld + k] * ref[(size_t)j * ld + k];
    ref[(size_t)j * ld + j] = sqrt(ref[(size_t)j * ld + j]);
    for (int i = j + 1; i < nrow; ++i) {
      double a = ref[(size_t)i * ld + j];
      for (int k = 0; k < j; ++k) a -= ref[(size_t)i * ld + k] * ref[(size_t)j * ld + k];
      ref[(size_t)i * ld + j] = a / ref[(size_t)j * ld + j];
    }
  }
  double e = 0;
  for (int i = 0; i < nrow; ++i)
    for (int j = 0; j < cw && j <= i; ++j) e = fmax(e, fabs(L[(size_t)i * ld + j] - ref[(size_t)i * ld + j]));
  printf("check: max|L - L_host| = %.2e\n", e);
#ifdef CHAIN_STAMPS
  std::vector<unsigned long long> st(64);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_chain_stamps), sizeof(unsigned long long) * 64);
  printf("k_chain_block (cycles): ");
  for (int p = 0; p < 4; ++p)
    printf("| p%d potrf %llu stage %llu solve %llu upd %llu put %llu ", p, st[1 + 5 * p] - st[5 * p], st[2 + 5 * p] - st[1 + 5 * p],
           st[3 + 5 * p] - st[2 + 5 * p], st[4 + 5 * p] - st[3 + 5 * p], st[5 + 5 * p] - st[4 + 5 * p]);
  printf("\nk_trsm_rows (cycles): entry->descriptors %llu, ->prologue issued %llu, ->first barrier %llu;", st[61] - st[60], st[32] - st[61], st[33] - st[32]);
  for (int t = 0; t < 10; ++t) printf(" [%d] product %llu, to next barrier %llu;", t, st[34 + 2 * t] - st[33 + 2 * t], t < 9 ? st[35 + 2 * t] - st[34 + 2 * t] : 0ull);
  printf("\n");
#endif
  return 0;
}
