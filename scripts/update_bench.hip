// Times k_update<T> alone on one synthetic DIRECT unit:  C[MxN] -= A[MxK] * B[NxK]^T
// with A, B rows of one block column of width K (as the trailing update).
// Build on the box:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -I include -I spllt_amd/csrc scripts/update_bench.hip -o /tmp/update_bench
#include "../spllt_amd/csrc/kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace spx;

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int K : {64, 128, 256, 512, 1024}) {
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
    u.a_off = 0; u.a_w = K;   // block column of segment 0 (carried by the unit)
    if (getenv("UB_ATOMIC")) u.atomic = 1;   // epilogue: atomic subtract instead of read-modify-write
    UpdUnit* du;
    hipMalloc(&du, sizeof(u));
    hipMemcpy(du, &u, sizeof(u), hipMemcpyHostToDevice);
    for (int T : {128, 64, 32}) {
      std::vector<UpdTile> tl;
      for (int tj = 0; tj < (N + T - 1) / T; ++tj)
        for (int ti = 0; ti < (M + T - 1) / T; ++ti) tl.push_back(UpdTile{0, (short)ti, (short)tj});
      UpdTile* dt;
      hipMalloc(&dt, tl.size() * sizeof(UpdTile));
      hipMemcpy(dt, tl.data(), tl.size() * sizeof(UpdTile), hipMemcpyHostToDevice);
      // SUSTAINED rate: the chip needs ~100 ms of continuous load to reach its clocks (the first
      // launches after an idle pause run 15-20 % slower, scripts/sustain_probe.hip); round 2 timed
      // single launches between synchronisations and read 62 TFLOP/s where the kernel holds 68
      float best = 1e9;
      {
        const double est_ms = 2.0 * M * N * K / 40e12 * 1e3 + 0.01;
        const int nwarm = (int)(150.0 / est_ms) + 1, nrep = (int)(60.0 / est_ms) + 3;
        for (int r = 0; r < nwarm; ++r)
          launch_update(0, T, dt, (int64_t)tl.size(), du, bc_off, bc_w, L, nullptr, nullptr, nullptr);
        hipEventRecord(e0);
        for (int r = 0; r < nrep; ++r)
          launch_update(0, T, dt, (int64_t)tl.size(), du, bc_off, bc_w, L, nullptr, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms / nrep;
      }
      // one lone tile: latency of a single workgroup
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch_update(0, T, dt, 1, du, bc_off, bc_w, L, nullptr, nullptr, nullptr);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms1;
      hipEventElapsedTime(&ms1, e0, e1);
      printf("K=%4d T=%3d tiles=%5zu  %8.1f us  %6.2f TFLOP/s   lone tile %.1f us\n", K, T, tl.size(),
             best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12, ms1 * 1e3);
      hipFree(dt);
    }
    hipFree(L); hipFree(bc_off); hipFree(bc_w); hipFree(du);
  }
  return 0;
}
