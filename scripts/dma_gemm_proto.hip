// Prototype: the update product C[MxN] -= A[MxK] B[NxK]^T (both operands K-contiguous) with the
// operand tiles loaded STRAIGHT INTO LDS (global_load_lds_dwordx4: no register staging, no
// ds_write) and two LDS stages (one barrier per K step), 128 x 128 tile, 8 waves (4 x 2), BK = 16.
// The 16-byte chunks of a row are stored XOR-swizzled by (row >> 1) & 7 so that the MFMA operand
// reads (16 rows x one k) are conflict-free without padding (the DMA writes whole 1 KB runs).
// Question it answers: how far above k_update<128> (59 TFLOP/s at K = 1024) does this get?
//   hipcc --offload-arch=gfx950 -O3 scripts/dma_gemm_proto.hip -o bin_tmp/dma_gemm_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int T = 128, BK = 16, WM = 4, WN = 2, NT = 64 * WM * WN;
constexpr int FMM = T / WM / 16, FMN = T / WN / 16;   // 2 x 4 fragments per wave
constexpr int ROWB = BK * 8;                          // bytes per tile row in LDS (128)

__device__ __forceinline__ void dma16(const double* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(NT, 2) void k_dma(const double* __restrict__ A, const double* __restrict__ B,
                                               double* __restrict__ C, int M, int N, int K, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [stage][A|B][T rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
  // DMA mapping: wave w, instruction i in {0, 1}: tile rows (2w + i) * 8 + (lane >> 3), chunk lane & 7
  const int lr8 = lane >> 3, lc = lane & 7;
  const double* ga[2];
  const double* gb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (2 * wave + i) * 8 + lr8;
    const int cg = lc ^ ((row >> 1) & 7);          // global chunk that lands in LDS chunk lc of this row
    ga[i] = A + (size_t)min(i0 + row, M - 1) * ld + 2 * cg;
    gb[i] = B + (size_t)min(j0 + row, N - 1) * ld + 2 * cg;
  }
  auto issue = [&](int st, int k) {
    char* base = smem + st * (2 * T * ROWB);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma16(ga[i] + k, base + (2 * wave + i) * 8 * ROWB);
      dma16(gb[i] + k, base + T * ROWB + (2 * wave + i) * 8 * ROWB);
    }
  };
  d4 acc[FMM][FMN];
#pragma unroll
  for (int a = 0; a < FMM; ++a)
#pragma unroll
    for (int b = 0; b < FMN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const int lr = lane & 15, lq = lane >> 4;
  // ragged K: the last chunk was loaded whole; the columns beyond K are cleared in LDS (once, in
  // the step that holds them) so that the MFMA loop needs no masks
  auto clear_tail = [&](int stg, int k) {
    const int rem = K - k;                 // valid columns of this step (< BK)
    char* base = smem + stg * (2 * T * ROWB);
    const int row = tid >> 1;              // 256 rows (A tile, then B tile), two threads per row
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = (tid & 1) * 8 + e;
      if (kk >= rem)
        *(double*)(base + row * ROWB + (((kk >> 1) ^ (((row & (T - 1)) >> 1) & 7)) << 4) + ((kk & 1) << 3)) = 0.0;
    }
  };
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (K < BK) { clear_tail(0, 0); __syncthreads(); }
  int st = 0;
  for (int k = 0; k < K; k += BK) {
    if (k + BK < K) issue(st ^ 1, k + BK);
    const char* As = smem + st * (2 * T * ROWB);
    const char* Bs = As + T * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double af[FMM], bf[FMN];
      const int kk = 4 * ks + lq, cg = kk >> 1, half = kk & 1;
#pragma unroll
      for (int a = 0; a < FMM; ++a) {
        const int row = wm * (T / WM) + a * 16 + lr;
        af[a] = *(const double*)(As + row * ROWB + ((cg ^ ((row >> 1) & 7)) << 4) + (half << 3));
      }
#pragma unroll
      for (int b = 0; b < FMN; ++b) {
        const int row = wn * (T / WN) + b * 16 + lr;
        bf[b] = *(const double*)(Bs + row * ROWB + ((cg ^ ((row >> 1) & 7)) << 4) + (half << 3));
      }
#pragma unroll
      for (int a = 0; a < FMM; ++a)
#pragma unroll
        for (int b = 0; b < FMN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k + BK < K && k + 2 * BK > K) {     // the stage just filled holds the ragged last step
      clear_tail(st ^ 1, k + BK);
      __syncthreads();
    }
    st ^= 1;
  }
#pragma unroll
  for (int a = 0; a < FMM; ++a) {
    double cv[4][FMN];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wm * (T / WM) + a * 16 + lq + 4 * r;
#pragma unroll
      for (int b = 0; b < FMN; ++b) {
        const int j = j0 + wn * (T / WN) + b * 16 + lr;
        cv[r][b] = C[(size_t)min(i, M - 1) * N + min(j, N - 1)];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wm * (T / WM) + a * 16 + lq + 4 * r;
#pragma unroll
      for (int b = 0; b < FMN; ++b) {
        const int j = j0 + wn * (T / WN) + b * 16 + lr;
        if (i < M && j < N) C[(size_t)i * N + j] = cv[r][b] - acc[a][b][r];
      }
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned lds = 2 * 2 * T * ROWB;
  hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int K : {64, 128, 250, 255, 256, 512, 1024}) {
    const int ld = (K == 250) ? 251 : K;   // (K = 250: odd row stride, rows only 8-byte aligned; 255: odd too)

    std::vector<double> ha((size_t)M * ld + 64), hb((size_t)N * ld + 64);
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = ((i * 40503u) % 1000) * 1e-3 - 0.5;
    double *A, *B, *C;
    hipMalloc(&A, ha.size() * 8); hipMalloc(&B, hb.size() * 8); hipMalloc(&C, (size_t)M * N * 8);
    hipMemcpy(A, ha.data(), ha.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
    hipMemset(C, 0, (size_t)M * N * 8);
    dim3 grid((N + T - 1) / T, (M + T - 1) / T);
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_dma, grid, dim3(NT), lds, 0, A, B, C, M, N, K, ld);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    // check one entry: after 6 launches C[i][j] = -6 * sum_k A[i][k] B[j][k]
    double c;
    const int ci = 77 % M, cj = 1234 % N;
    hipMemcpy(&c, C + (size_t)ci * N + cj, 8, hipMemcpyDeviceToHost);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += ha[(size_t)ci * ld + k] * hb[(size_t)cj * ld + k];
    printf("K=%5d  %8.1f us  %6.2f TFLOP/s   check %.3e\n", K, best * 1e3, 2.0 * M * N * K / best / 1e9,
           fabs(c + 6 * ref));
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
