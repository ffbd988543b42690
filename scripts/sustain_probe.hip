// Does the update kernel slow down under sustained load?  The 128-tile at K = 1024 (M = N = 8192),
// launched back to back for about a second after an idle pause: rate of every one of the first
// launches, then averages over groups of 50.  (rocBLAS DGEMM reaches 72 TFLOP/s on these shapes in
// a process of its own and 64 inside the bench process, where this kernel does 62-64: is that the
// chip's power management or the kernels?)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -I include -I spllt_amd/csrc scripts/sustain_probe.hip -o bin_tmp/sustain_probe
#include "../spllt_amd/csrc/kernels.hip"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
using namespace spx;

int main(int argc, char** argv) {
  const int M = 8192, N = 8192, K = argc > 1 ? atoi(argv[1]) : 1024, T = argc > 2 ? atoi(argv[2]) : 128;
  const int NL = argc > 3 ? atoi(argv[3]) : 400;
  const int64_t src_elems = (int64_t)(M + N) * K, dst_elems = (int64_t)M * N;
  double* L;
  hipMalloc(&L, (src_elems + dst_elems + 64) * 8);
  std::vector<double> h(src_elems);
  for (int64_t i = 0; i < src_elems; ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  hipMemcpy(L, h.data(), src_elems * 8, hipMemcpyHostToDevice);
  hipMemset(L + src_elems, 0, dst_elems * 8);
  int64_t bc_off_h[2] = {0, src_elems};
  int bc_w_h[2] = {K, N};
  int64_t* bc_off;
  int* bc_w;
  hipMalloc(&bc_off, 16);
  hipMalloc(&bc_w, 8);
  hipMemcpy(bc_off, bc_off_h, 16, hipMemcpyHostToDevice);
  hipMemcpy(bc_w, bc_w_h, 8, hipMemcpyHostToDevice);
  UpdUnit u{};
  u.d_off = src_elems; u.src_bcol0 = 0; u.nseg = 1; u.seg_r0 = 0; u.seg_stride = K;
  u.src_r0 = N; u.src_c0 = 0; u.M = M; u.N = N; u.k0 = 0; u.klen = -1; u.d_ld = N;
  u.d_row0 = 0; u.d_col0 = 0; u.mode = MODE_DIRECT; u.lower = 0; u.b_bcol0 = -1;
  u.a_off = 0; u.a_w = K;
  UpdUnit* du;
  hipMalloc(&du, sizeof(u));
  hipMemcpy(du, &u, sizeof(u), hipMemcpyHostToDevice);
  std::vector<UpdTile> tl;
  for (int tj = 0; tj < (N + T - 1) / T; ++tj)
    for (int ti = 0; ti < (M + T - 1) / T; ++ti) tl.push_back(UpdTile{0, (short)ti, (short)tj});
  UpdTile* dt;
  hipMalloc(&dt, tl.size() * sizeof(UpdTile));
  hipMemcpy(dt, tl.data(), tl.size() * sizeof(UpdTile), hipMemcpyHostToDevice);
  std::vector<hipEvent_t> ev((size_t)NL + 1);
  for (auto& e : ev) hipEventCreate(&e);
  // one launch to load the code object, then the chip idles
  launch_update(0, T, dt, (int64_t)tl.size(), du, bc_off, bc_w, L, nullptr, nullptr, nullptr);
  hipDeviceSynchronize();
  for (int round = 0; round < 2; ++round) {
    std::this_thread::sleep_for(std::chrono::milliseconds(round == 0 ? 3000 : 200));
    hipEventRecord(ev[0]);
    for (int i = 0; i < NL; ++i) {
      launch_update(0, T, dt, (int64_t)tl.size(), du, bc_off, bc_w, L, nullptr, nullptr, nullptr);
      hipEventRecord(ev[(size_t)i + 1]);
    }
    hipDeviceSynchronize();
    const double fl = 2.0 * M * N * K;
    printf("round %d (after %s idle): K=%d T=%d, %d launches back to back\n  first launches (TFLOP/s):", round,
           round == 0 ? "3 s" : "0.2 s", K, T, NL);
    float tot = 0;
    for (int i = 0; i < NL; ++i) {
      float ms;
      hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]);
      if (i < 12) printf(" %.1f", fl / (ms * 1e-3) / 1e12);
      tot += ms;
      if ((i + 1) % 50 == 0) {
        if (i + 1 == 50) printf("\n  groups of 50:");
        float g;
        hipEventElapsedTime(&g, ev[(size_t)i + 1 - 50], ev[(size_t)i + 1]);
        printf(" %.1f", 50 * fl / (g * 1e-3) / 1e12);
      }
    }
    printf("\n  all: %.1f TFLOP/s over %.0f ms\n", NL * fl / (tot * 1e-3) / 1e12, tot);
  }
  return 0;
}
