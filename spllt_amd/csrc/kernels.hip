// Hand-written CDNA4 (gfx950) kernels of the factorize hot path.
//
//   k_scatter_val   a8  spllt_init_node   (reference src/spllt_kernels_mod.F90:2301-2364)
//   k_init_arena    a8  the same with the clearing of the arena in the same pass (single GPU, eager)
//   k_chain_potrf   a11 spllt_factor_diag_block (:1168-1189) on one <=64-wide panel per
//                   workgroup, also emits the inverse of the factored panel (the default
//                   chain step); k_potrf_panel: the same body for the operator twins
//   k_panel         a whole panel step of a block column with few rows in one launch: POTRF,
//                   the rows below (a12) and the left-looking update of the next panel (a13)
//   k_update<T>     a12 spllt_solve_block (:1217) as X = A * inv(L_pp)^T,
//                   a13 spllt_update_block (:1261-1292),
//                   a16+a18 spllt_update_between + spllt_expand_buffer
//                   (:2108-2237, :2010-2053) with the scatter fused into the GEMM
//                   epilogue, or (deterministic engine) stored into a scratch block
//                   -- one fp64-MFMA kernel, four epilogues.
//   k_update_dma128 the 128-tile of the same product with the operand tiles DMA'd straight into
//                   two XOR-swizzled LDS stages (global_load_lds_dwordx4)
//   k_gather        a18 turned destination-centric: ordered assembly of the buffered update
//                   blocks (deterministic engine, no atomics)
//   k_scatter_block a26 spllt_scatter_block (:1122-1160) extend-add
//   k_solve_diag / k_solve_strip   forward / backward substitution on the device-resident
//                   factor (reference src/spllt_solve_mod.F90), up to 4 right-hand sides
//   k_flag_pack / k_flag_unpack   multi-GPU: not-positive-definite indicator <-> exchange buffer
//   k_poison_lds    debug: fills the LDS of every CU with signalling-NaN patterns
//
// Storage convention: every block column of L is a
// row-major (rows x width) matrix; all products are C = A * B^T with both
// operands K-contiguous, which is exactly the operand order
// v_mfma_f64_16x16x4_f64 wants: lane l supplies A[l&15][l>>4] and B[l>>4][l&15]
// and receives C[(l>>4) + 4r][l&15] in register r.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <tuple>
#include <type_traits>

#include "kernels.hpp"

// shape of the 128-tile update kernel (scripts/update_bench.hip builds variants with
// -DUPD128_BK / -DUPD128_WM / -DUPD128_WN)
#ifndef UPD128_BK
#define UPD128_BK 16
#endif
#ifndef UPD128_WM
#define UPD128_WM 4
#define UPD128_WN 2
#endif

// (experiment switch: waves per SIMD the 64-tile update kernel is compiled for; 0 = as many as its
// ~90 registers allow (5))
#ifndef UPD64_OCC
#define UPD64_OCC 0
#endif

// (experiment switch: what the scatter epilogue would cost without atomics -- results are WRONG
// with it, timing only)
#if defined(SCATTER_EXPERIMENT) && SCATTER_EXPERIMENT == 1
#define SCATTER_ADD(p, v) (*(p) += (v))
#elif defined(SCATTER_EXPERIMENT) && SCATTER_EXPERIMENT == 2
#define SCATTER_ADD(p, v) (*(p) = (v))
#elif defined(SCATTER_EXPERIMENT) && SCATTER_EXPERIMENT == 3
#define SCATTER_ADD(p, v) ((void)(p), (void)(v))
#else
#define SCATTER_ADD(p, v) unsafeAtomicAdd((p), (v))
#endif

namespace spx {

typedef double d4 __attribute__((ext_vector_type(4)));

// launch, or add as a kernel node (LaunchSink)
template <class... KArgs, class... Args>
static void emit(const LaunchSink& s, void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, Args... args) {
  if (!s.graph) {
    hipLaunchKernelGGL(kernel, grid, block, lds, s.stream, args...);
    return;
  }
  std::tuple<KArgs...> vals(static_cast<KArgs>(args)...);
  void* params[sizeof...(KArgs)];
  size_t i = 0;
  std::apply([&](auto&... v) { ((params[i++] = (void*)&v), ...); }, vals);
  hipKernelNodeParams kp{};
  kp.func = reinterpret_cast<void*>(kernel);
  kp.gridDim = grid;
  kp.blockDim = block;
  kp.sharedMemBytes = lds;
  kp.kernelParams = params;
  kp.extra = nullptr;
  s.err = hipGraphAddKernelNode(&s.node, s.graph, s.deps, s.ndeps, &kp);
}

// ---------------------------------------------------------------------------
// a8: L[dst[i]] = val[src[i]]  (assignment; the arena was zeroed beforehand)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_val(double* __restrict__ L,
                                                     const double* __restrict__ val,
                                                     const int64_t* __restrict__ dst,
                                                     const int64_t* __restrict__ src, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) L[dst[i]] = val[src[i]];
}

// ---------------------------------------------------------------------------
// a11: Cholesky of one <=64 x <=64 diagonal panel block per workgroup plus
// X = L^-1 (lower triangular, row-major ld = n, strictly-upper part zero)
// written to the dinv scratch.  The block lives in LDS as 4x4 sub-blocks of
// 16x16:
//   * off-diagonal work (block-column updates, panel solves through the
//     inverted 16x16 diagonal blocks, and the block recurrences of the inverse)
//     runs on v_mfma_f64_16x16x4_f64, one sub-block per wavefront;
//   * the 16x16 diagonal blocks are factored AND inverted in registers by one
//     wavefront in the same loop (lane i owns row i of D and column i of inv(D);
//     the pivot column is broadcast with v_readlane, 1/sqrt from v_rsq_f64 + Newton).
// An accumulator tile S (C layout: reg r = row (l>>4)+4r, col l&15) is exactly
// the B operand of k-step r, so X_IJ = -inv(D_I) * S needs no data movement.
// A non-positive pivot records (pivot column + 1) in *flag (smallest wins).
// ---------------------------------------------------------------------------
constexpr int TLD = 66;   // LDS row stride (doubles): conflict-free MFMA operand reads
constexpr int DLD = 17;

template <int LD>
__device__ inline d4 ld_c_s(const double* M, int row0, int col0, int lane) {
  const int lq = lane >> 4, lr = lane & 15;
  d4 c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = M[(row0 + lq + 4 * r) * LD + col0 + lr];
  return c;
}
template <int LD>
__device__ inline void st_c_s(double* M, int row0, int col0, int lane, d4 c) {
  const int lq = lane >> 4, lr = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) M[(row0 + lq + 4 * r) * LD + col0 + lr] = c[r];
}
__device__ inline d4 ld_c(const double* M, int row0, int col0, int lane) {
  const int lq = lane >> 4, lr = lane & 15;
  d4 c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = M[(row0 + lq + 4 * r) * TLD + col0 + lr];
  return c;
}
__device__ inline void st_c(double* M, int row0, int col0, int lane, d4 c) {
  const int lq = lane >> 4, lr = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) M[(row0 + lq + 4 * r) * TLD + col0 + lr] = c[r];
}

// broadcast of lane K of every row of 16 lanes to the lanes of that row, in the vector ALU
// (DPP row_newbcast: no trip through the scalar registers, no SALU/VALU hazard waits).  The
// register Cholesky keeps four identical copies of its 16 x 16 block, one per row of lanes, so
// this is the broadcast of lane K of the wave.
template <int K>
__device__ __forceinline__ double bcast_row(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// (src is a compile-time constant once the caller's loops are unrolled: the switch folds away)
__device__ __forceinline__ double bcast16(double v, int src) {
  switch (src) {
    case 0: return bcast_row<0>(v);
    case 1: return bcast_row<1>(v);
    case 2: return bcast_row<2>(v);
    case 3: return bcast_row<3>(v);
    case 4: return bcast_row<4>(v);
    case 5: return bcast_row<5>(v);
    case 6: return bcast_row<6>(v);
    case 7: return bcast_row<7>(v);
    case 8: return bcast_row<8>(v);
    case 9: return bcast_row<9>(v);
    case 10: return bcast_row<10>(v);
    case 11: return bcast_row<11>(v);
    case 12: return bcast_row<12>(v);
    case 13: return bcast_row<13>(v);
    case 14: return bcast_row<14>(v);
    default: return bcast_row<15>(v);
  }
}
// broadcast of lane `src` (compile-time constant after unrolling) through SGPRs
__device__ inline double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// 1/sqrt(d) and sqrt(d) to fp64 accuracy from v_rsq_f64 + Newton steps
__device__ inline void rsqrt_sqrt(double d, double& y, double& r) {
  y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  r = d * y;
  r = __builtin_fma(0.5 * y, __builtin_fma(-r, r, d), r);
}

#ifdef POTRF_STAMPS
__device__ unsigned long long g_potrf_stamps[32];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_potrf_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP2(c, i) do { if ((c) && blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_potrf_stamps[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP2(c, i) do { } while (0)
#endif

#ifdef CHAIN_STAMPS
__device__ unsigned long long g_chain_stamps[64];
#define CSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_chain_stamps[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif

struct PotrfShared {
  double X[64 * TLD];
  double DI[4][16 * DLD];
  double RI[64];  // reciprocals of the diagonal of L
  double T[64 * TLD];  // last: k_panel reuses it (and the dynamic LDS behind it) for its row blocks
};

// Factor (and invert) one <=64 x <=64 block held at A (row stride ld); the whole
// workgroup takes part.  D receives inv(L) (row-major, row stride ldd).
__device__ __forceinline__ void potrf64_body(PotrfShared& sh, double* __restrict__ A, int ld, int n,
                                             double* __restrict__ D, int ldd, int gcol, int flags,
                                             int* __restrict__ flag) {
  double (&T)[64 * TLD] = sh.T;
  double (&X)[64 * TLD] = sh.X;
  double (&DI)[4][16 * DLD] = sh.DI;
  double (&RI)[64] = sh.RI;
  struct { int n, ld, gcol, flags; } u = {n, ld, gcol, flags};
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int nblk = (n + 15) >> 4;
  const int np = nblk * 16;
  const bool do_chol = !(u.flags & 1);
  STAMP(0);
  // identity-padded lower triangle into LDS.  Lane l of wave w takes column l of the rows
  // w, w+4, w+8, ...: one wave-instruction reads one whole row (4 cache lines; the earlier
  // mapping -- 16 consecutive columns per thread -- touched 64 lines per instruction and spent
  // 7.8 k of the kernel's 55 k cycles loading, 7.5 k storing).
  // (callers may run more than 256 threads: only the first 256 load / store / compute here,
  // the others just take part in the barriers)
  const int lw = tid >> 6, lc = tid & 63;
  if (tid < 256) {
    // unconditional loads at clamped addresses (all 16 in flight at once; a load under a
    // condition is compiled into a branch with its own wait), selection afterwards
    double v[16];
    const int cc = lc < n ? lc : n - 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = 4 * e + lw;
      v[e] = A[(int64_t)(r < n ? r : n - 1) * ld + cc];
    }
    // all 64 rows are written: the matrix-core steps of the callers read whole
    // 16-row fragments of X, and stale LDS may hold NaN bit patterns
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = 4 * e + lw;
      const double x = (r < n && lc <= r) ? v[e] : ((r == lc) ? 1.0 : 0.0);
      T[r * TLD + lc] = x;
      X[r * TLD + lc] = 0.0;
      if (!do_chol && lc == r) RI[r] = 1.0 / x;
    }
  }
  __syncthreads();
  STAMP(1);
  for (int J = 0; do_chol && J < nblk; ++J) {
    // A1: T[I][J] -= sum_{K<J} T[I][K] T[J][K]^T, one sub-block per wave
    if (J > 0) {
      const int I = J + w;
      if (I < nblk) {
        d4 acc = ld_c(T, I * 16, J * 16, lane);
        for (int K = 0; K < J; ++K) {
          double a[4], b[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            a[t] = -T[(I * 16 + lr) * TLD + K * 16 + 4 * t + lq];
            b[t] = T[(J * 16 + lr) * TLD + K * 16 + 4 * t + lq];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
        }
        st_c(T, I * 16, J * 16, lane, acc);
      }
      __syncthreads();
    }
    STAMP(2 + 3 * J);
    // A2: factor the diagonal 16x16 block in registers (wave 0: lane i owns
    // row i; the four 16-lane groups hold identical copies, group 0 writes)
    if (w == 0) {
      double row[16];
      // inv(D_J) rides along: lane c owns COLUMN c of W = inv(D_J); its forward
      // substitution consumes the very scalars the factorization broadcasts
      // (column j of D_J and 1/L_jj), so it only adds 16-j FMAs per step
      double wacc[16], wx[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        row[c] = T[(J * 16 + lr) * TLD + J * 16 + c];
        wacc[c] = 0.0;
      }
      int failcol = 1 << 30;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        // unscaled column j of the other rows: independent of the pivot's
        // reciprocal square root, so the readlanes overlap its latency
        double tk[16];
#pragma unroll
        for (int k = j + 1; k < 16; ++k) tk[k] = bcast16(row[j], k);
        double djj = bcast16(row[j], j);
        if (!(djj > 0.0)) {
          if (failcol == (1 << 30)) failcol = j;
          djj = 1.0;
        }
        double y, d;
        rsqrt_sqrt(djj, y, d);
        const double sc = row[j] * (y * y);   // L_ij * y  (= row[j] * y^2)
        // unconditional: entries above the diagonal (lr < k) become garbage
        // that nothing reads (the write-back masks them)
        // W[j][c] = (delta_jc - sum_{k<j} L_jk W[k][c]) / L_jj ;  L_kj = tk[k] * y
        wx[j] = (((lr == j) ? 1.0 : 0.0) - wacc[j]) * y;
        const double z = wx[j] * y;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
          row[k] -= sc * tk[k];
          wacc[k] = __builtin_fma(tk[k], z, wacc[k]);   // same broadcast scalar, used twice
        }
        row[j] = (lr == j) ? d : row[j] * y;
        RI[J * 16 + j] = y;  // wave-uniform value, every lane stores the same word
        // pin W[j][.] here: it is only stored by lanes < 16 below, and without the
        // pin the whole inverse recurrence is sunk into that branch, which keeps
        // every broadcast scalar of every step alive (SGPR spills)
        asm volatile("" : "+v"(wx[j]));
      }
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          T[(J * 16 + lr) * TLD + J * 16 + c] = (c <= lr) ? row[c] : 0.0;
          DI[J][c * DLD + lr] = wx[c];
          X[(J * 16 + c) * TLD + J * 16 + lr] = wx[c];
        }
        if (failcol != (1 << 30) && lane == 0 && !(u.flags & 4)) atomicMin(flag, u.gcol + J * 16 + failcol + 1);
      }
    }
    __syncthreads();
    STAMP(3 + 3 * J);
    // A3: sub-blocks below: X_IJ = A_IJ * inv(D_J)^T on the matrix core
    {
      const int I = J + 1 + w;
      if (I < nblk) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double a = T[(I * 16 + lr) * TLD + J * 16 + 4 * t + lq];
          const double b = DI[J][lr * DLD + 4 * t + lq];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        st_c(T, I * 16, J * 16, lane, acc);
      }
    }
    __syncthreads();
    STAMP(4 + 3 * J);
  }
  // B0 (invert-only blocks): invert the diagonal 16x16 blocks, one per wave: lane c
  // owns COLUMN c of inv(D_w) and solves D_w x = e_c by forward substitution; the
  // entries of D_w are wave-uniform LDS reads, so no cross-lane traffic is needed
  if (!do_chol && w < nblk) {
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double sacc = (i == lr) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) sacc -= T[(w * 16 + i) * TLD + w * 16 + k] * x[k];
      x[i] = (i >= lr) ? sacc * RI[w * 16 + i] : 0.0;
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        DI[w][i * DLD + lr] = x[i];
        X[(w * 16 + i) * TLD + w * 16 + lr] = x[i];
      }
    }
  }
  __syncthreads();
  STAMP(14);
  // B: X_IJ = -inv(D_I) * sum_{K=J}^{I-1} L_IK X_KJ, by block diagonals
  for (int d = 1; d < nblk; ++d) {
    const int J = w, I = J + d;
    if (I < nblk) {
      d4 S = {0.0, 0.0, 0.0, 0.0};
      for (int K = J; K < I; ++K) {
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          const double a = T[(I * 16 + lr) * TLD + K * 16 + k + lq];
          const double b = X[(K * 16 + k + lq) * TLD + J * 16 + lr];
          S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, S, 0, 0, 0);
        }
      }
      d4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double a = -DI[I][lr * DLD + 4 * t + lq];
        R = __builtin_amdgcn_mfma_f64_16x16x4f64(a, S[t], R, 0, 0, 0);
      }
      st_c(X, I * 16, J * 16, lane, R);
    }
    __syncthreads();
  }
  STAMP(15);
  // flags bit 1: the caller stores L itself; bit 2: redundant copy, neither the inverse is
  // stored nor a failed pivot reported
  if (tid < 256 && lc < n) {
    const bool st_l = do_chol && !(u.flags & 2), st_i = !(u.flags & 4);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = 4 * e + lw;
      if (r < n) {
        if (st_l && lc <= r) A[(int64_t)r * ld + lc] = T[r * TLD + lc];
        if (st_i) D[(int64_t)r * ldd + lc] = X[r * TLD + lc];
      }
    }
  }
  STAMP(16);
  (void)np;
}

// ---------------------------------------------------------------------------
// The default POTRF body (round 3).  Same result as potrf64_body (L_pp in place, inv(L_pp) to
// D), organised around ONE critical path -- the four 16 x 16 register Choleskys of wave 0 --
// with everything else moved into its shadow:
//   * the 16 x 16 factorization runs in LDL^T order: the serial chain per pivot is
//     d_j -> 1/d_j (v_rcp_f64 + two Newton steps) -> multiplier -> next diagonal entry, about
//     a third of the reciprocal-square-root chain; the square roots are taken once, for all 16
//     pivots at the same time (lane j: d_j), after the loop.  The broadcasts of the pivot column
//     cost nothing: v_fmac_f64_dpp row_newbcast reads lane k of the row of 16 lanes as an
//     operand (round 2: two v_mov_b32_dpp per FMA; 415 cycles per pivot, now ~110).
//   * right-looking: after the solve of the blocks below (A3) wave 0 only updates the NEXT
//     diagonal block and goes on; waves 1-3 meanwhile apply the finished block column to the
//     other trailing blocks, compute the finished block ROW of the inverse
//     (X_JK = -W_JJ sum_M L_JM X_MK: needs rows < J of X only) and send the finished rows of L
//     and X home (a CU stores ~10 B/cycle: 64 KB at the end of the kernel were 4-7 k cycles).
// flags: bit 1 the caller stores L itself; bit 2 neither the inverse is stored nor a failed pivot
// reported; bit 3 the upper triangle of the inverse's slot is already zero and stays so (the
// engine clears the dinv scratch once): only the lower triangle is stored; bit 4 the block is
// already in sh.T (identity-padded lower triangle), nothing is loaded.
// ---------------------------------------------------------------------------
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
// The register Cholesky below is a fixed sequence of volatile inline-assembly instructions (the
// compiler only allocates the registers): left to itself the compiler puts the whole reciprocal
// chain of pivot j+1 BEHIND the ~30 update FMAs of pivot j instead of into their issue slots, and
// sched_barrier does not bind inline assembly.  The hazards the compiler would pad are padded by
// hand and checked at build time (scripts/check_dpp_hazards.py):
//   * a VGPR written by a vector instruction must not be read by a DPP instruction in the next
//     two issue slots (s_nop 1 behind the pivot column's update, whose result is broadcast next);
//   * the result of v_rcp_f64 must not be consumed by the very next instruction (s_nop 0).
// acc += (lane K of this row of 16 lanes: s0) * s1, one VOP2-DPP instruction
template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double s0, double s1) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s0), "v"(s1), "n"(K));
}
// the same for the next pivot's column, followed by the broadcast of the new pivot d = acc[lane K]
template <int K>
__device__ __forceinline__ void fmac_bc_pivot(double& acc, double s0, double s1, double& d) {
  asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
               "v_mov_b64_dpp %1, %0 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
               : "+v"(acc), "=&v"(d) : "v"(s0), "v"(s1), "n"(K));
}
__device__ __forceinline__ double rcp_newton(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  return y;
}
__device__ __forceinline__ double a_rcp(double d) {
  double y;
  asm volatile("v_rcp_f64 %0, %1\n\ts_nop 0" : "=v"(y) : "v"(d));
  return y;
}
__device__ __forceinline__ double a_nr_err(double d, double y) {      // 1 - d y
  double e;
  asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(d), "v"(y));
  return e;
}
__device__ __forceinline__ void a_nr_fix(double& y, double e) {        // y += y e
  asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(y) : "v"(e));
}
__device__ __forceinline__ double a_nmul(double a, double b) {         // -(a b)
  double o;
  asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
__device__ __forceinline__ double a_mul(double a, double b) {
  double o;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
__device__ __forceinline__ double a_sub(double a, double b) {
  double o;
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(o) : "v"(a), "v"(b));
  return o;
}
// 16 x 16 Cholesky + inverse in registers.  Lane i of every row of 16 lanes holds row i of the
// block (row[c]; four identical copies per wave); on return row[c] = L[i][c] for c <= i
// (garbage above the diagonal) and wx[c] = W[c][i], column i of W = inv(L).  Returns the first
// failed pivot (1 << 30: none); a failed pivot is not repaired (the block fills with NaNs or
// garbage and the factorization is reported as failed).
__device__ __forceinline__ int chol16_regs(double (&row)[16], double (&wx)[16], int lr) {
  double wacc[16], delta[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    wacc[c] = 0.0;
    delta[c] = (lr == c) ? 1.0 : 0.0;
  }
  double dn = bcast_row<0>(row[0]);
  double dmine = dn;
  double y = rcp_newton(dn);
  STAMP2(true, 21);
  static_for<0, 16>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    using std::integral_constant;
    // row i, column j holds m_ij = l_ij d_j (unscaled); multiplier l_ij = m_ij / d_j = m_ij y
    const double nsc = a_nmul(row[j], y);
    if constexpr (j + 1 < 16) {
      // the updates of pivot j -- a_ik -= l_ij m_kj (k = j+2..15), then the inverse of the unit
      // lower factor that rides along, wacc[k] += l_kj Wu[j][c] (k = j+1..15) -- are dealt into
      // the latency gaps of pivot j+1's reciprocal chain
      constexpr int nR = 14 - j, nF = 29 - 2 * j;
      double wxj = 0.0, z = 0.0;
      auto fill = [&](auto lo, auto hi) {
        static_for<decltype(lo)::value, decltype(hi)::value>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr (i < nR) {
            fmac_bc<j + 2 + i>(row[j + 2 + i], row[j], nsc);
          } else {
            if constexpr (i == nR) {
              // Wu[j][c] = delta_jc - sum_{k<j} l_jk Wu[k][c]
              wxj = a_sub(delta[j], wacc[j]);
              z = a_mul(wxj, y);
            }
            fmac_bc<j + 1 + i - nR>(wacc[j + 1 + i - nR], row[j], z);
          }
        });
      };
      constexpr int b0 = nF * 30 / 100, b1 = nF * 44 / 100, b2 = nF * 58 / 100, b3 = nF * 72 / 100;
      fmac_bc_pivot<j + 1>(row[j + 1], row[j], nsc, dn);   // the next pivot's column first
      double yn = a_rcp(dn);
      fill(integral_constant<int, 0>{}, integral_constant<int, b0>{});
      double e = a_nr_err(dn, yn);
      fill(integral_constant<int, b0>{}, integral_constant<int, b1>{});
      a_nr_fix(yn, e);
      fill(integral_constant<int, b1>{}, integral_constant<int, b2>{});
      e = a_nr_err(dn, yn);
      fill(integral_constant<int, b2>{}, integral_constant<int, b3>{});
      a_nr_fix(yn, e);
      fill(integral_constant<int, b3>{}, integral_constant<int, nF>{});
      dmine = (lr == j + 1) ? dn : dmine;
      wx[j] = wxj;
      y = yn;
    } else {
      wx[j] = delta[j] - wacc[j];
    }
  });
  STAMP2(true, 22);
  // L = M diag(d)^-1/2, W = diag(d)^-1/2 Wu: all 16 square roots at once (lane j: d_j)
  double ys, rs;
  rsqrt_sqrt(dmine, ys, rs);
  // (the diagonal comes out as d_j / sqrt(d_j): within an ulp or two of sqrt(d_j); the pins
  // keep the products here: the callers store row[] and wx[] in different branches, into which
  // the compiler would otherwise sink them one by one)
  static_for<0, 16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double sc = bcast_row<c>(ys);
    row[c] *= sc;
    wx[c] *= sc;
    asm volatile("" : "+v"(row[c]), "+v"(wx[c]));
  });
  (void)rs;
  STAMP2(true, 23);
  // lane j holds pivot j: the first one that is not positive
  const unsigned long long bad = __ballot(!(dmine > 0.0)) & 0xffffull;
  return bad ? __builtin_ctzll(bad) : (1 << 30);
}

__device__ __forceinline__ void potrf64_v2(PotrfShared& sh, double* __restrict__ A, int ld, int n,
                                           double* __restrict__ D, int ldd, int gcol, int flags,
                                           int* __restrict__ flag) {
  double (&T)[64 * TLD] = sh.T;
  double (&X)[64 * TLD] = sh.X;
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int nblk = (n + 15) >> 4;
  const bool st_l = !(flags & 2), st_i = !(flags & 4), x_lower = (flags & 8) != 0;
  STAMP(0);
  // identity-padded lower triangle into LDS (see potrf64_body)
  const int lw = tid >> 6, lc = tid & 63;
  if (flags & 16) {
    // the caller has put the (identity-padded, lower) block into sh.T itself: only X is cleared
    if (tid < 256) {
#pragma unroll
      for (int e = 0; e < 16; ++e) X[(4 * e + lw) * TLD + lc] = 0.0;
    }
  } else
  if (tid < 256) {
    // (unconditional loads at clamped addresses; the columns right of the row's diagonal block
    // re-read its last column, which costs no traffic)
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = 4 * e + lw;
      const int cmax = min(n - 1, r | 15);
      v[e] = A[(int64_t)(r < n ? r : n - 1) * ld + (lc < cmax ? lc : cmax)];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) X[(4 * e + lw) * TLD + lc] = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = 4 * e + lw;
      T[r * TLD + lc] = (r < n && lc <= r) ? v[e] : ((r == lc) ? 1.0 : 0.0);
    }
  }
  __syncthreads();
  STAMP(1);
  // rows [row0, row0 + 8) x columns [col0, col0 + 16) of the LDS image M (row stride TLD) -> dst
  // (row stride ldst), entries on or below the diagonal only: ONE wave-instruction, 8 lanes per
  // row, two columns (16 bytes) per lane -- a CU retires stores by the instruction, and one row of
  // 8-byte lanes per instruction made the stores the longest part of the kernel
  auto store_rows8 = [&](double* __restrict__ dst, int ldst, const double* M, int row0, int col0) {
    const int g = row0 + (lane >> 3), c = col0 + 2 * (lane & 7);
    const int cend = min(g, n - 1);                 // last column to store
    if (g >= n || c > cend) return;
    const double v0 = M[g * TLD + c], v1 = M[g * TLD + c + 1];
    double* q = dst + (int64_t)g * ldst + c;
    if (c + 1 <= cend) {
      typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
      *reinterpret_cast<d2*>(q) = (d2){v0, v1};
    } else {
      q[0] = v0;
    }
  };
  // block (I, Jp) -= T[I][S] T[Jp][S]^T
  auto upd_block = [&](int I, int Jp, int S) {
    d4 acc = ld_c(T, I * 16, Jp * 16, lane);
    double a[4], b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = -T[(I * 16 + lr) * TLD + S * 16 + 4 * t + lq];
      b[t] = T[(Jp * 16 + lr) * TLD + S * 16 + 4 * t + lq];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
    st_c(T, I * 16, Jp * 16, lane, acc);
  };
  // The work in the shadow of wave 0, shared by ns waves (this one: ws); step S is final: block
  // column S of L and W_SS.  Wave ws owns block column K = ws of the inverse: Xpre carries
  // sum_{M=K}^{S-1} L_SM X_MK from the previous shadow (it needs nothing of step S), so that
  // X_SK = -W_SS Xpre is four MFMAs once W_SS exists -- which matters for the LAST block row,
  // the only one that is not hidden behind a register Cholesky.
  d4 Xpre = {0.0, 0.0, 0.0, 0.0};
  auto shadow = [&](int S, int ws, int ns, bool stores) {
    const int K = ws;
    if (K < S) {
      d4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const double a = -X[(S * 16 + lr) * TLD + S * 16 + 4 * tt + lq];     // W_SS[lr][k]
        R = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xpre[tt], R, 0, 0, 0);
      }
      st_c(X, S * 16, K * 16, lane, R);
      if (st_i && x_lower) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = S * 16 + lq + 4 * r;
          if (g < n) D[(int64_t)g * ldd + K * 16 + lr] = R[r];
        }
      }
    }
    int t = 0;
    for (int Jp = S + 1; Jp < nblk; ++Jp)
      for (int I = Jp; I < nblk; ++I) {
        if (I == S + 1 && Jp == S + 1) continue;         // wave 0's
        if (t++ % ns == ws) upd_block(I, Jp, S);
      }
    if (S + 1 < nblk && K <= S) {
      // for the next block row: sum_{M=K}^{S} L_{S+1,M} X_MK  (X_SK: just written by this wave, or W_SS)
      Xpre = (d4){0.0, 0.0, 0.0, 0.0};
      for (int M = K; M <= S; ++M) {
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          const double a = T[((S + 1) * 16 + lr) * TLD + M * 16 + k + lq];
          const double b = X[(M * 16 + k + lq) * TLD + K * 16 + lr];
          Xpre = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, Xpre, 0, 0, 0);
        }
      }
    }
    // block column S of L (final with step S) and the diagonal block of the inverse go home
    if (!stores) return;
    const int ngrp = (64 - S * 16) / 8;
    for (int gi = ws; gi < ngrp; gi += ns)
      if (st_l) store_rows8(A, ld, T, S * 16 + gi * 8, S * 16);
    if (st_i && x_lower)
      for (int gi = ws; gi < 2; gi += ns) store_rows8(D, ldd, X, S * 16 + gi * 8, S * 16);
  };
  for (int J = 0; J < nblk; ++J) {
    if (w == 0) {
      if (J > 0) upd_block(J, J, J - 1);                  // the only update on the critical path
      double row[16], wx[16];
      STAMP2(true, 20);
#pragma unroll
      for (int c = 0; c < 16; ++c) row[c] = T[(J * 16 + lr) * TLD + J * 16 + c];
      const int failcol = chol16_regs(row, wx, lr);
      STAMP2(true, 24);
      // the four rows of 16 lanes hold identical copies: row 0 writes L_JJ (the entries above
      // the diagonal are never read by anybody: left as they are), row 1 W_JJ into the inverse
      // (where the solve of the blocks below and the recurrences of the inverse read it)
      if (lq == 0) {
        double* dst = &T[(J * 16 + lr) * TLD + J * 16];
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c] = row[c];
        if (failcol != (1 << 30) && lane == 0 && st_i) atomicMin(flag, gcol + J * 16 + failcol + 1);
      } else if (lq == 1) {
        double* dst = &X[(J * 16) * TLD + J * 16 + lr];
#pragma unroll
        for (int c = 0; c < 16; ++c) dst[c * TLD] = wx[c];
      }
      STAMP2(true, 25);
    } else if (w < 4 && J > 0) {
      shadow(J - 1, w - 1, 3, true);
    }
    __syncthreads();
    STAMP(2 + 2 * J);
    // blocks below: X_IJ = A_IJ inv(D_J)^T
    if (w < 4 && J + 1 + w < nblk) {
      const int I = J + 1 + w;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double a = T[(I * 16 + lr) * TLD + J * 16 + 4 * t + lq];
        const double b = X[(J * 16 + lr) * TLD + J * 16 + 4 * t + lq];     // W_JJ[lr][k]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      st_c(T, I * 16, J * 16, lane, acc);
    }
    __syncthreads();
    STAMP(3 + 2 * J);
  }
  // the last step's shadow: the last block row of the inverse, the last diagonal blocks (wave 0
  // takes the stores of its own blocks, waves 1-3 finish their block of the inverse)
  if (w > 0 && w < 4) {
    const int S = nblk - 1, K = w - 1;
    if (K < S) shadow(S, K, 3, false);                    // (no updates are left; wave 0 stores)
  } else if (w == 0) {
    const int S = nblk - 1;
    if (st_l) { store_rows8(A, ld, T, S * 16, S * 16); store_rows8(A, ld, T, S * 16 + 8, S * 16); }
    if (st_i && x_lower) { store_rows8(D, ldd, X, S * 16, S * 16); store_rows8(D, ldd, X, S * 16 + 8, S * 16); }
  }
  STAMP(10);
  if (st_i && !x_lower) {
    // a caller-owned slot that is not known to be zero above the diagonal (operator twins): whole rows
    __syncthreads();
    if (tid < 256)
      for (int r = lw; r < n; r += 4)
        if (lc < n) D[(int64_t)r * ldd + lc] = X[r * TLD + lc];
  }
  STAMP(11);
}

// factorizations take the round-3 body, "invert only" (flags bit 0: operator twins) the older one
__device__ __forceinline__ void potrf64(PotrfShared& sh, double* __restrict__ A, int ld, int n,
                                        double* __restrict__ D, int ldd, int gcol, int flags,
                                        int* __restrict__ flag) {
#ifndef POTRF_V1
  if (!(flags & 1)) {
    potrf64_v2(sh, A, ld, n, D, ldd, gcol, flags, flag);
    return;
  }
#endif
  potrf64_body(sh, A, ld, n, D, ldd, gcol, flags, flag);
}

// (u0 = units[0] by value: the descriptor of workgroup 0 -- on the critical path of the top
// levels a launch has ONE workgroup -- arrives with the kernel arguments instead of through a
// dependent global load in front of the block's loads)
__global__ __launch_bounds__(256) void k_potrf_panel(const PotrfUnit* __restrict__ units,
                                                     double* __restrict__ L,
                                                     double* __restrict__ dinv,
                                                     int* __restrict__ flag, const PotrfUnit u0) {
  __shared__ PotrfShared sh;
  __builtin_amdgcn_s_setprio(3);
  const PotrfUnit u = blockIdx.x == 0 ? u0 : units[blockIdx.x];
  potrf64(sh, L + u.off, u.ld, u.n, dinv + u.dinv_off, u.n, u.gcol, u.flags, flag);
}

// ---------------------------------------------------------------------------
// One step of the panel chain (ChainUnit), one workgroup per block column: panel [c0, c0+pn)
// of a block column (a11 spllt_factor_diag_block): L_pp = chol(A_pp) and inv(L_pp) into the
// dinv scratch, which turns the triangular solve of the rows below into a product
// (k_update, TRSM mode) and is what the solve phase applies.  256 threads and static LDS: two
// workgroups per CU -- the leaf levels of a large problem are thousands of such panels and run
// at the rate the chip retires these workgroups.
// (Round 2 also had a 768-thread k_chain_panel that walked diagonal sub-tiles wider than a
// panel -- solve and update of the sub-tile's rows inside the chain kernel, a k_winv kernel
// for the left part of the rows below -- selected by a "chain block" knob.  It was slower at
// every setting and every one of the three hangs this repository has
// seen on the GPU box happened in a case that used it; it has been removed.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chain_potrf(const ChainUnit* __restrict__ units,
                                                     double* __restrict__ L,
                                                     double* __restrict__ dinv,
                                                     int* __restrict__ flag, const ChainUnit u0) {
  __shared__ PotrfShared sh;
  __builtin_amdgcn_s_setprio(3);
  const ChainUnit u = blockIdx.x == 0 ? u0 : units[blockIdx.x];
  const int cq = u.c0 - u.cs;
  // (flags 8: the dinv scratch is cleared when it is allocated, and nothing ever writes above the
  // diagonal of a slot; the inverse's rows have the stride of their chain block, schedule.hpp winv_ld)
  potrf64(sh, L + u.off + (int64_t)u.c0 * u.ld + u.c0, u.ld, u.pn, dinv + u.winv_off + cq, u.ce - u.cs,
          u.gcol, 8, flag);
}

// ---------------------------------------------------------------------------
// One whole panel step of a block column in ONE launch (PanelUnit, tiles = 64-row blocks of
// the rows below the panel), for the latency-bound levels of the tree: every workgroup
//   1. factors the panel's diagonal block itself (redundantly: ~24 us that would otherwise
//      be a launch of its own),
//   2. solves its own rows:            X_i = A[i, panel] inv(L_pp)^T          (a12)
//   3. solves the NEXT panel's pivot rows the same way (redundantly, in LDS),
//   4. applies the left-looking update of the next panel's columns to its own rows:
//      A[i, next] -= [X_i(:, 0:c0) | X_i] [X_d(:, 0:c0) | X_d]^T              (a13)
// so that per panel one kernel boundary is on the critical path instead of three (POTRF,
// TRSM, in-panel update).  No workgroup waits for another one: everything a workgroup reads
// was final before the launch or is recomputed by itself.  The two blocks that several
// workgroups READ and that the step OVERWRITES -- the diagonal block (becomes L_pp) and the
// next pivot rows of the panel's columns (become X_d) -- are written by whichever workgroup
// read them LAST (a counter per unit; all workgroups hold bit-identical results), so a
// workgroup that starts late (more workgroups than free CUs) still finds them unsolved.
// LDS: PotrfShared (T is reused for X_i once the factor is done, X for staging once the
// inverse is no longer needed) + two 64 x TLD buffers behind it (dynamic).
// ---------------------------------------------------------------------------
constexpr int kPanelThreads = 512;

// 64 x 64 block at src (row stride ld) -> dst[64][TLD]; rows >= nrow / columns >= ncol zeroed
// (unconditional loads at clamped addresses, selection afterwards; 8 lanes read 64 contiguous bytes)
__device__ __forceinline__ void stage_block(double* __restrict__ dst, const double* __restrict__ src,
                                            int64_t ld, int nrow, int ncol, int tid) {
  const int r = tid >> 3, cl = tid & 7;
  const double* row = src + (int64_t)(r < nrow ? r : nrow - 1) * ld;
  double v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = row[cl + 8 * e < ncol ? cl + 8 * e : ncol - 1];
#pragma unroll
  for (int e = 0; e < 8; ++e) dst[r * TLD + cl + 8 * e] = (r < nrow && cl + 8 * e < ncol) ? v[e] : 0.0;
}

// out[16 s + .][16 jb + .] (two column blocks jb0, jb0+1) += sum_k a[row][k] b[col][k] over 64 k
__device__ __forceinline__ void mma_64(const double* __restrict__ a, const double* __restrict__ b, int s,
                                       int jb0, int lane, d4& acc0, d4& acc1) {
  const int lq = lane >> 4, lr = lane & 15;
  double av[16], b0[16], b1[16];
#pragma unroll
  for (int kt = 0; kt < 16; ++kt) {
    av[kt] = a[(s * 16 + lr) * TLD + 4 * kt + lq];
    b0[kt] = b[(jb0 * 16 + lr) * TLD + 4 * kt + lq];
    b1[kt] = b[((jb0 + 1) * 16 + lr) * TLD + 4 * kt + lq];
  }
#pragma unroll
  for (int kt = 0; kt < 16; ++kt) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kt], b0[kt], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kt], b1[kt], acc1, 0, 0, 0);
  }
}

// the same product with the operand fragments requested in two halves (48 instead of 96 registers
// in flight: for kernels that hold other blocks in registers meanwhile)
__device__ __forceinline__ void mma_64_lean(const double* __restrict__ a, const double* __restrict__ b, int s,
                                            int jb0, int lane, d4& acc0, d4& acc1) {
  const int lq = lane >> 4, lr = lane & 15;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    double av[8], b0[8], b1[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      av[kt] = a[(s * 16 + lr) * TLD + 4 * (8 * hf + kt) + lq];
      b0[kt] = b[(jb0 * 16 + lr) * TLD + 4 * (8 * hf + kt) + lq];
      b1[kt] = b[((jb0 + 1) * 16 + lr) * TLD + 4 * (8 * hf + kt) + lq];
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kt], b0[kt], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kt], b1[kt], acc1, 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
}

__device__ __forceinline__ void mma_64x4(const double* __restrict__ a, const double* __restrict__ b, int s,
                                         int lane, d4 (&acc)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  mma_64(a, b, s, 0, lane, acc[0], acc[1]);
  mma_64(a, b, s, 2, lane, acc[2], acc[3]);
}

// "I have read the shared block": true for the workgroup that says so last (it also clears the
// counter for the next factorization).  Called by all threads after a barrier that follows the
// reads; the result is workgroup-uniform.
__device__ __forceinline__ bool last_reader(int* __restrict__ counter, int readers, int* __restrict__ slot) {
  if (threadIdx.x == 0) {
    __threadfence();
    const int old = atomicAdd(counter, 1);
    const bool last = old == readers - 1;
    if (last) atomicExch(counter, 0);
    *slot = last ? 1 : 0;
  }
  __syncthreads();
  const bool last = *slot != 0;
  __syncthreads();
  return last;
}

__global__ __launch_bounds__(kPanelThreads) void k_panel(const UpdTile* __restrict__ tiles,
                                                         const PanelUnit* __restrict__ units,
                                                         double* __restrict__ L,
                                                         double* __restrict__ dinv,
                                                         int* __restrict__ counters,
                                                         int* __restrict__ flag) {
  extern __shared__ __attribute__((aligned(16))) double panel_smem[];
  __shared__ int vote;
  PotrfShared& sh = *reinterpret_cast<PotrfShared*>(panel_smem);
  double* Ui = sh.T;                                          // X_i (T is dead after the factorization)
  double* Ud = panel_smem + sizeof(PotrfShared) / sizeof(double);  // X_d
  double* S = Ud + 64 * TLD;                                  // staging
  __builtin_amdgcn_s_setprio(3);
  const UpdTile tl = tiles[blockIdx.x];
  const PanelUnit u = units[tl.unit];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int ld = u.ld, c0 = u.c0, pn = u.pn;
  double* A = L + u.off;
  const bool first = tl.ti == 0;
  // ---- 1. the panel's diagonal block (the first workgroup stores the inverse and reports a
  // failed pivot; L_pp goes home from the workgroup that read the block last) ---------------
  double* Dg = A + (int64_t)c0 * ld + c0;
  const bool factored = (u.pad_ & 1) != 0;
  if (factored) {
    // the panel was factored by a chain launch: only its inverse is needed here (the lower
    // triangle comes from the dinv scratch, the rest of the 64 x 64 image is zero)
    const double* W = dinv + u.dinv_off;
    for (int e = tid; e < 64 * 64; e += kPanelThreads) {
      const int r = e >> 6, c = e & 63;
      sh.X[r * TLD + c] = (r < pn && c <= r) ? W[(int64_t)r * pn + c] : 0.0;
    }
  } else {
    potrf64(sh, Dg, ld, pn, dinv + u.dinv_off, pn, u.gcol, first ? (2 | 8) : 6, flag);
  }
  __syncthreads();
  if (!factored && last_reader(counters + 2 * tl.unit, u.ntile, &vote)) {
    const int lc = tid & 63;       // a wave-instruction writes (part of) one row
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * e + (tid >> 6);
      if (r < pn && lc <= r) Dg[(int64_t)r * ld + lc] = sh.T[r * TLD + lc];
    }
    __syncthreads();
  }
  const int r1 = c0 + pn;                  // first row below the panel
  const int r0 = r1 + 64 * (int)tl.ti;     // first row of this workgroup's block
  const int nr = min(64, u.nrow - r0);     // (<= 0: the block column ends with the panel)
  if (nr <= 0) return;
  const int pn2 = u.next_pn;
  const int s = wave >> 1, jb0 = (wave & 1) * 2;   // this wave's 16-row strip and two 16-column blocks
  // ---- 2. X_i = A[r0.., panel] inv(L_pp)^T ---------------------------------------------
  stage_block(S, A + (int64_t)r0 * ld + c0, ld, nr, pn, tid);
  __syncthreads();
  // the next panel's pivot rows are the first pn2 rows of the first block: every other
  // workgroup reads them unsolved in step 3
  bool store_pivot = pn2 > 0 && first && last_reader(counters + 2 * tl.unit + 1, u.ntile, &vote);
  {
    d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
    mma_64(S, sh.X, s, jb0, lane, a0, a1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = s * 16 + lq + 4 * r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = (jb0 + h) * 16 + lr;
        const double v = (j < pn) ? (h ? a1[r] : a0[r]) : 0.0;   // padding columns: exact zeros
        Ui[i * TLD + j] = v;
        const bool mine = !first || i >= pn2 || store_pivot;
        if (mine && i < nr && j < pn) A[(int64_t)(r0 + i) * ld + c0 + j] = v;
      }
    }
  }
  if (pn2 <= 0) return;                    // last panel of the block column: nothing to update
  __syncthreads();                         // S and X are free again; Ui is complete
  // ---- 3. X_d: the next panel's pivot rows, solved the same way ---------------------------
  const double* Xd = Ui;
  if (!first) {
    stage_block(S, A + (int64_t)r1 * ld + c0, ld, pn2, pn, tid);
    __syncthreads();
    store_pivot = last_reader(counters + 2 * tl.unit + 1, u.ntile, &vote);
    d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
    mma_64(S, sh.X, s, jb0, lane, a0, a1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = s * 16 + lq + 4 * r;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = (jb0 + h) * 16 + lr;
        const double v = (j < pn) ? (h ? a1[r] : a0[r]) : 0.0;
        Ud[i * TLD + j] = v;
        if (store_pivot && i < pn2 && j < pn) A[(int64_t)(r1 + i) * ld + c0 + j] = v;
      }
    }
    Xd = Ud;
    __syncthreads();
  }
  // ---- 4. A[r0.., next panel] -= [X_i(:, 0:c0) | X_i] [X_d(:, 0:c0) | X_d]^T (lower part) ----
  const int cn = r1;                       // first column of the next panel (= its first pivot row)
  d4 c0v, c1v;
  bool ok0[4], ok1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = s * 16 + lq + 4 * r;
    const int j0 = jb0 * 16 + lr, j1 = j0 + 16;
    ok0[r] = i < nr && j0 < pn2 && (r0 + i) >= (cn + j0);
    ok1[r] = i < nr && j1 < pn2 && (r0 + i) >= (cn + j1);
    const double* crow = A + (int64_t)(r0 + (i < nr ? i : nr - 1)) * ld + cn;
    c0v[r] = crow[j0 < pn2 ? j0 : pn2 - 1];       // unconditional, clamped
    c1v[r] = crow[j1 < pn2 ? j1 : pn2 - 1];
  }
  d4 m0 = {0.0, 0.0, 0.0, 0.0}, m1 = {0.0, 0.0, 0.0, 0.0};
  double* Sa = sh.X;                        // the inverse is no longer needed: staging for the own rows
  {
    // K in 64-column chunks through LDS; the loads of chunk k + 1 fly during the MFMAs of chunk k
    // (one chunk after the other, loads and MFMAs in turn, was 3-6 us per chunk: a whole global
    // round trip each, on the critical path of every fused panel step)
    const int sr = tid >> 3, scl = tid & 7;
    const double* rowa = A + (int64_t)(r0 + (sr < nr ? sr : nr - 1)) * ld;
    const double* rowd = A + (int64_t)(r1 + (sr < pn2 ? sr : pn2 - 1)) * ld;
    double va[8], vd[8];
    auto fetch = [&](int k0) {
      const int kw = min(64, c0 - k0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = k0 + (scl + 8 * e < kw ? scl + 8 * e : kw - 1);
        va[e] = rowa[c];
        vd[e] = rowd[c];
      }
    };
    if (c0 > 0) fetch(0);
    for (int k0 = 0; k0 < c0; k0 += 64) {
      const int kw = min(64, c0 - k0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool in = scl + 8 * e < kw;
        Sa[sr * TLD + scl + 8 * e] = (sr < nr && in) ? va[e] : 0.0;
        S[sr * TLD + scl + 8 * e] = (sr < pn2 && in) ? vd[e] : 0.0;
      }
      __syncthreads();
      if (k0 + 64 < c0) fetch(k0 + 64);
      mma_64(Sa, S, s, jb0, lane, m0, m1);
      __syncthreads();
    }
  }
  mma_64(Ui, Xd, s, jb0, lane, m0, m1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = s * 16 + lq + 4 * r;
    double* crow = A + (int64_t)(r0 + (i < nr ? i : nr - 1)) * ld + cn;
    if (ok0[r]) crow[jb0 * 16 + lr] = c0v[r] - m0[r];
    if (ok1[r]) crow[jb0 * 16 + 16 + lr] = c1v[r] - m1[r];
  }
}

// ---------------------------------------------------------------------------
// A whole small subtree in ONE workgroup (SubTask; L_SUBTREE): the reference's subtree task
// (a20-a25: spllt_subtree_factorize, src/spllt_factorization_mod.F90:196-261; its kernels
// kernels_mod:97-821).  The nodes of the task are walked in post-order; every node has one block
// column of at most one panel:
//   1. L_ss = chol(A_ss), W = inv(L_ss)                       (potrf64: a11)
//   2. X = A_below W^T, 64 rows at a time                      (a12)
//   3. its update units, one 64 x 64 tile X_I X_J^T at a time (a16-a19):
//        MODE_SCATTER  into a node of the same subtree (only this workgroup touches it before the
//                      launch ends) or, from the ROOT, into the ancestors above the subtree
//        MODE_GEN      what leaves the subtree from a node below the root: into the subtree's
//                      generated element (packed lower triangle over the root's rows below its
//                      columns, a few hundred KB: cache-resident while the subtree is worked on)
//      the root's tiles take the generated element along: dest -= X_I X_J^T + G_IJ, G_IJ = 0 --
//      the ONE extend-add of the subtree (factorization_mod:39-191); the scratch is zero again when
//      the launch ends.
// All adds are atomics without return (the workgroup does not wait for them); a fence + barrier
// per node makes them visible to the loads of the nodes that follow.
// LDS: PotrfShared only (T = staging of A / X_I, X = W, then X_J): two workgroups per CU.
// ---------------------------------------------------------------------------
constexpr int kSubThreads = 512;

__global__ __launch_bounds__(kSubThreads) void k_subtree(const SubTask* __restrict__ tasks,
                                                         const SubNode* __restrict__ nodes,
                                                         const UpdUnit* __restrict__ units,
                                                         const int* __restrict__ relpos,
                                                         const int* __restrict__ rlist,
                                                         double* __restrict__ L,
                                                         double* __restrict__ dinv,
                                                         double* __restrict__ G,
                                                         int* __restrict__ flag) {
  __shared__ PotrfShared sh;
  const SubTask T = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int s = wave >> 1, jb0 = (wave & 1) * 2;   // this wave's 16-row strip and two 16-column blocks
  for (int ni = 0; ni < T.node_count; ++ni) {
    const SubNode nd = nodes[T.node_first + ni];
    double* A = L + nd.off;
    const int w = nd.w, m = nd.nrow, ld = nd.w;
    // ---- 1. the diagonal block ----------------------------------------------------------
    potrf64(sh, A, ld, w, dinv + nd.dinv_off, w, nd.gcol, 8, flag);
    __syncthreads();
    // ---- 2. the rows below ---------------------------------------------------------------
    for (int r0 = w; r0 < m; r0 += 64) {
      const int nr = min(64, m - r0);
      stage_block(sh.T, A + (int64_t)r0 * ld, ld, nr, w, tid);
      __syncthreads();
      d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
      mma_64(sh.T, sh.X, s, jb0, lane, a0, a1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = s * 16 + lq + 4 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = (jb0 + h) * 16 + lr;
          if (i < nr && j < w) A[(int64_t)(r0 + i) * ld + j] = h ? a1[r] : a0[r];
        }
      }
      __syncthreads();
    }
    __threadfence();
    __syncthreads();
    // ---- 3. the update units ---------------------------------------------------------------
    const bool with_gen = nd.root != 0 && T.g_n > 0;
    for (int ui = 0; ui < nd.unit_count; ++ui) {
      const UpdUnit u = units[nd.unit_first + ui];
      const bool gen = u.mode == MODE_GEN;
      const int nti = (u.M + 63) >> 6, ntj = (u.N + 63) >> 6;
      double* dst = (gen ? G : L) + u.d_off;
      for (int tj = 0; tj < ntj; ++tj) {
        const int c0 = u.src_c0 + 64 * tj, nc = min(64, u.N - 64 * tj);
        stage_block(sh.X, A + (int64_t)c0 * ld, ld, nc, w, tid);
        // the columns of this wave's entries in the destination
        int dcol[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = (jb0 + h) * 16 + lr;
          const int64_t at = u.gcol_off + 64 * tj + (j < nc ? j : nc - 1);
          dcol[h] = gen ? relpos[at] : rlist[at] - u.d_col0;
        }
        for (int ti = 0; ti < nti; ++ti) {
          const int r0 = u.src_r0 + 64 * ti, nr = min(64, u.M - 64 * ti);
          if (r0 + nr - 1 < c0) continue;                  // entirely above the diagonal
          const double* Sa = sh.X;
          if (r0 != c0 || nr != nc) {                      // (the diagonal tile of a square corner: X_I = X_J)
            stage_block(sh.T, A + (int64_t)r0 * ld, ld, nr, w, tid);
            Sa = sh.T;
          }
          int drow[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = s * 16 + lq + 4 * r;
            drow[r] = relpos[u.relrow_off + 64 * ti + (i < nr ? i : nr - 1)] - (gen ? 0 : u.d_row0);
          }
          __syncthreads();
          d4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
          mma_64(Sa, sh.X, s, jb0, lane, a0, a1);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = s * 16 + lq + 4 * r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int j = (jb0 + h) * 16 + lr;
              if (i >= nr || j >= nc || r0 + i < c0 + j) continue;
              double v = h ? a1[r] : a0[r];
              if (gen) {
                unsafeAtomicAdd(dst + (int64_t)drow[r] * (drow[r] + 1) / 2 + dcol[h], -v);
              } else {
                if (with_gen) {
                  const int gi = r0 + i - w, gj = c0 + j - w;
                  double* q = G + T.g_off + (int64_t)gi * (gi + 1) / 2 + gj;
                  v -= *q;
                  *q = 0.0;
                }
                unsafeAtomicAdd(dst + (int64_t)drow[r] * u.d_ld + dcol[h], -v);
              }
            }
          }
          __syncthreads();
        }
        __syncthreads();
      }
    }
    __threadfence();
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// One step of the panel chain for a CHAIN BLOCK of up to four panels (ChainUnit with pn = its
// width cw <= 4 pw), one workgroup: the whole cw x cw diagonal block of a block column is
// factored here, panel by panel, right-looking, with nothing but this workgroup's barriers
// between the steps (a11 spllt_factor_diag_block, kernels_mod:1168-1189, on the diagonal block):
//   for p = 0 .. np-1:   L_pp = chol(A_pp), W_pp = inv(L_pp)      (potrf64, waves 0-3)
//                        L_ip = A_ip W_pp^T            i > p     (64^3 products, all waves)
//                        A_ij -= L_ip L_jp^T           i >= j > p
// The blocks below the current panel are staged in LDS (at most three: two buffers behind
// PotrfShared and sh.T, which is free between two factorizations); the trailing blocks stay
// where they are -- the arena, 512 KB at most, in L2 -- read and written by this workgroup only
// (its waves share the CU's vector L1; a barrier behind the stores orders them).  The inverses
// stay per panel (W_pp: what k_trsm_rows and the solve phase read).
// Why: per 256 columns the chain stream carried 4 x (POTRF, TRSM, in-panel update) = 12 dependent
// launches; with this kernel and k_trsm_rows it carries two.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kPanelThreads) void k_chain_block(const ChainUnit* __restrict__ units,
                                                               double* __restrict__ L,
                                                               double* __restrict__ dinv,
                                                               int* __restrict__ flag, int pw,
                                                               const ChainUnit u0) {
  extern __shared__ __attribute__((aligned(16))) double cblk_smem[];
  PotrfShared& sh = *reinterpret_cast<PotrfShared*>(cblk_smem);
  __builtin_amdgcn_s_setprio(3);
  const ChainUnit u = blockIdx.x == 0 ? u0 : units[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int s = wave >> 1, jb0 = (wave & 1) * 2;   // this wave's 16-row strip and two 16-column blocks
  const int cw = u.pn, ld = u.ld;
  const int np = (cw + pw - 1) / pw;
  double* A = L + u.off + (int64_t)u.c0 * ld + u.c0;   // entry (0, 0) of the diagonal block
  double* W = dinv + u.winv_off;                        // slot of the block's first panel; panel p: + p pw^2
  // (offsets, not pointers: a select between LDS pointers goes through generic pointers, and this
  // compiler then emits a compare against src_shared_base that its own verifier rejects)
  constexpr int kPaOff = (int)(sizeof(PotrfShared) / sizeof(double));
  constexpr int kTOff = (int)(offsetof(PotrfShared, T) / sizeof(double));
  auto pbuf = [&](int q) { return cblk_smem + (q == 0 ? kPaOff : (q == 1 ? kPaOff + 64 * TLD : kTOff)); };
  CSTAMP(0);
  for (int p = 0; p < np; ++p) {
    const int n_p = min(pw, cw - p * pw);
    // (from the second panel on, the diagonal block is already in sh.T: the update phase below
    // put it there instead of sending it through the arena)
    potrf64(sh, A + (int64_t)(p * pw) * ld + p * pw, ld, n_p, W + (int64_t)p * pw * pw, n_p, u.gcol + p * pw,
            p == 0 ? 8 : (8 | 16), flag);
    __syncthreads();                                    // (the tail of potrf64 still read sh.T / sh.X)
    CSTAMP(1 + 5 * p);
    const int nbel = np - 1 - p;                        // blocks below this panel (<= 3)
    if (nbel == 0) break;
    if (p == 0) {
#pragma unroll 1
      for (int q = 0; q < nbel; ++q) {
        const int i = 1 + q, n_i = min(pw, cw - i * pw);
        stage_block(pbuf(q), A + (int64_t)(i * pw) * ld, ld, n_i, n_p, tid);
      }
      __syncthreads();
    }
    // the destination entries of a trailing block in accumulator layout (clamped, unconditional)
    auto cload = [&](int a, int b, d4& v0, d4& v1) {
      const int i = p + 1 + a, j = p + 1 + b;
      const int n_i = min(pw, cw - i * pw), n_j = min(pw, cw - j * pw);
      const double* C = A + (int64_t)(i * pw) * ld + j * pw;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = s * 16 + lq + 4 * r;
        const double* crow = C + (int64_t)(row < n_i ? row : n_i - 1) * ld;
        const int ca = jb0 * 16 + lr, cb2 = ca + 16;
        v0[r] = crow[ca < n_j ? ca : n_j - 1];
        v1[r] = crow[cb2 < n_j ? cb2 : n_j - 1];
      }
    };
    CSTAMP(2 + 5 * p);
    d4 c0v, c1v, n0v = {0.0, 0.0, 0.0, 0.0}, n1v = n0v;
    cload(0, 0, c0v, c1v);                              // flies during the solves
    // ---- L_ip = A_ip W_pp^T, into LDS (operands of the updates) and home ------------------
    // (runtime loops, one block at a time: unrolled, the operand fragments of all blocks are
    // requested at once and the kernel spills)
#pragma unroll 1
    for (int q = 0; q < nbel; ++q) {
      const int i = p + 1 + q, n_i = min(pw, cw - i * pw);
      double* Pq = pbuf(q);
      d4 x0 = {0.0, 0.0, 0.0, 0.0}, x1 = {0.0, 0.0, 0.0, 0.0};
      mma_64_lean(Pq, sh.X, s, jb0, lane, x0, x1);
      __syncthreads();                                  // everybody has read the A_ip image
      double* dst = A + (int64_t)(i * pw) * ld + p * pw;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = s * 16 + lq + 4 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = (jb0 + h) * 16 + lr;
          const double v = (row < n_i && col < n_p) ? (h ? x1[r] : x0[r]) : 0.0;   // padding: exact zeros
          Pq[row * TLD + col] = v;
          if (row < n_i && col < n_p) dst[(int64_t)row * ld + col] = v;
        }
      }
    }
    __syncthreads();
    CSTAMP(3 + 5 * p);
    // ---- A_ij -= L_ip L_jp^T, i >= j > p: the destination entries of the next block are
    // requested before the product of the current one.  The blocks of the NEXT panel's column
    // (b == 0) do not go back to the arena: the diagonal one becomes the input of the next
    // factorization in sh.T, the ones below it the staged operands of the next solves.
    d4 k0[3], k1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) k0[a] = k1[a] = (d4){0.0, 0.0, 0.0, 0.0};
    {
      int a = 0, b = 0;
      const int nupd = nbel * (nbel + 1) / 2;
#pragma unroll 1
      for (int t = 0; t < nupd; ++t) {
        // (the blocks row by row: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2))
        const int a2 = b < a ? a : a + 1, b2 = b < a ? b + 1 : 0;
        if (t + 1 < nupd) cload(a2, b2, n0v, n1v);
        const int i = p + 1 + a, j = p + 1 + b;
        const int n_i = min(pw, cw - i * pw), n_j = min(pw, cw - j * pw);
        d4 m0 = {0.0, 0.0, 0.0, 0.0}, m1 = {0.0, 0.0, 0.0, 0.0};
        mma_64_lean(pbuf(a), pbuf(b), s, jb0, lane, m0, m1);
        if (b == 0) {
#pragma unroll
          for (int aa = 0; aa < 3; ++aa)
            if (aa == a) {
              k0[aa] = c0v - m0;
              k1[aa] = c1v - m1;
            }
        } else {
          double* C = A + (int64_t)(i * pw) * ld + j * pw;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = s * 16 + lq + 4 * r;
            const int ca = jb0 * 16 + lr, cb2 = ca + 16;
            if (row >= n_i) continue;
            // (diagonal blocks: the lower triangle only -- what lies above it is never written)
            if (ca < n_j && (a != b || ca <= row)) C[(int64_t)row * ld + ca] = c0v[r] - m0[r];
            if (cb2 < n_j && (a != b || cb2 <= row)) C[(int64_t)row * ld + cb2] = c1v[r] - m1[r];
          }
        }
        c0v = n0v;
        c1v = n1v;
        a = a2;
        b = b2;
      }
    }
    __syncthreads();                                    // everybody has read the L_ip images
    CSTAMP(4 + 5 * p);
    {
      const int n_n = min(pw, cw - (p + 1) * pw);      // the next panel
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = s * 16 + lq + 4 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = (jb0 + h) * 16 + lr;
          // identity-padded lower triangle, the image potrf64 builds for itself (flag 16)
          sh.T[row * TLD + col] = (row < n_n && col <= row) ? (h ? k1[0][r] : k0[0][r]) : (row == col ? 1.0 : 0.0);
#pragma unroll
          for (int aa = 1; aa < 3; ++aa) {
            if (aa >= nbel) continue;
            const int n_i = min(pw, cw - (p + 1 + aa) * pw);
            pbuf(aa - 1)[row * TLD + col] = (row < n_i && col < n_n) ? (h ? k1[aa][r] : k0[aa][r]) : 0.0;
          }
        }
      }
    }
    // (potrf64 begins with a barrier of its own behind clearing sh.X)
    CSTAMP(5 + 5 * p);
  }
}

// ---------------------------------------------------------------------------
// The rows below a chain block, solved against its factored diagonal block (a12 spllt_solve_block,
// kernels_mod:1217-1229, for all panels of the block at once): one workgroup per 64 rows and ALL
// cw <= 4 pw columns of the block (UpdTile: unit, ti; the unit a TRSM-mode UpdUnit with N = cw),
//   X_p = (A_p - sum_{q<p} X_q L_pq^T) W_pp^T,   p = 0 .. np-1,
// computed TRANSPOSED, Y_p = X_p^T = W_pp (A_p^T - sum_q L_pq Y_q): the blocks of the diagonal
// factor (L_pq, W_pp: written by k_chain_block) are the MFMA A operands, staged through LDS for
// the whole workgroup, and a wave keeps the Y of its 16 rows in registers, where an accumulator
// tile (register r of lane l: row (l >> 4) + 4 r, column l & 15) IS the B operand of k-step r of
// the next product -- nothing of X goes through LDS.  Within every group of 16 columns the
// accumulator row (l >> 4) + 4 r stands for column 4 (l >> 4) + r (perm4 below), so that a lane
// holds four consecutive columns of a row (32-byte loads / stores); the staged operand blocks
// are permuted the same way in both directions, which keeps the operand reads on the
// conflict-free pattern of mma_64.  LDS: two operand stages (68 KB): two workgroups per CU.
// In place: a workgroup reads its rows before it writes them and nobody else touches them.
// Replaces, per 256 columns, four TRSM launches (K = 64: 151 launches at 3 TFLOP/s on the bench
// workload) and three in-panel update launches on the chain stream.
// ---------------------------------------------------------------------------
constexpr int kTrsmRows = 64;
__device__ __forceinline__ int perm4(int x) { return (x & ~15) | ((x & 3) << 2) | ((x >> 2) & 3); }

__global__ __launch_bounds__(256, 2) void k_trsm_rows(const UpdTile* __restrict__ tiles,
                                                      const UpdUnit* __restrict__ units,
                                                      double* __restrict__ L,
                                                      const double* __restrict__ dinv, int pw, int prio) {
  extern __shared__ __attribute__((aligned(16))) double trows_smem[];
  double* Bst = trows_smem;                         // [2][64][TLD]: L_pq / W_pp, permuted
  if (prio) __builtin_amdgcn_s_setprio(3);
  CSTAMP(60);
  const UpdTile tl = tiles[blockIdx.x];
  const UpdUnit u = units[tl.unit];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int cw = u.N, ld = u.d_ld;
  const int np = (cw + pw - 1) / pw;
  const int r0 = (int)tl.ti * kTrsmRows, nr = min(kTrsmRows, u.M - r0);
  double* X = L + u.d_off + (int64_t)(u.d_row0 + r0) * ld + u.d_col0;
  const double* Ld = L + u.d_off + (int64_t)u.d_col0 * ld + u.d_col0;    // the factored diagonal block
  const double* W = dinv + u.dinv_off;
  CSTAMP(61);
  // this lane's row of the block and its four consecutive columns 16 i + 4 lq + (0..3) of panel pp
  const int myrow = wave * 16 + lr;
  const bool row_ok = myrow < nr;
  double* Xrow = X + (int64_t)(row_ok ? myrow : nr - 1) * ld;
  auto aload = [&](int pp, d4 (&v)[4]) {
    const int n_pp = min(pw, cw - pp * pw);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * i + 4 * lq + r;
        v[i][r] = Xrow[pp * pw + (col < n_pp ? col : n_pp - 1)];
      }
  };
  // Operand blocks (64 x 64, 16 registers per thread each) are requested kDepth products ahead.
  // The blocks in the order they are used, (p, q): q < p is L_pq, q == p is W_pp.
  constexpr int kDepth = 1;
  constexpr int PT[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
  constexpr int QT[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
  const int nT = np * (np + 1) / 2;
  const int fr = tid >> 3, fc = tid & 7;
  double bv[kDepth][16];
  auto blk_rows = [&](int pp) { return min(pw, cw - pp * pw); };
  auto blk_cols = [&](int pp, int qq) { return qq == pp ? min(pw, cw - pp * pw) : pw; };
  auto fetch = [&](int pp, int qq, double (&v)[16]) {
    const int nrow = blk_rows(pp), ncol = blk_cols(pp, qq);
    const double* src = qq == pp ? W + (int64_t)pp * pw * pw : Ld + (int64_t)(pp * pw) * ld + qq * pw;
    const int64_t l = qq == pp ? nrow : ld;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = fr + 32 * h;
      const double* row = src + (int64_t)(r < nrow ? r : nrow - 1) * l;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[8 * h + e] = row[fc + 8 * e < ncol ? fc + 8 * e : ncol - 1];
    }
  };
  d4 ac[4];                   // A_p^T of the next panel to start (accumulator layout), requested a step ahead
  aload(0, ac);
#pragma unroll
  for (int t = 0; t < kDepth; ++t)
    if (t < nT) fetch(PT[t], QT[t], bv[t]);
  d4 Yn[3][4];                // -Y_0 .. -Y_2: this wave's 16 rows, all 64 columns of a panel each
  d4 g[4];                    // A_p^T - sum_q L_pq Y_q
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = (d4){0.0, 0.0, 0.0, 0.0};
  CSTAMP(32);
  // acc[i] += sum_k Ablk[16 i + .][k] Bf[k / 16][.]   (Ablk: a permuted LDS image, Bf: four accumulator
  // tiles); the operand fragments of k-step r + 1 are read while the MFMAs of step r issue
  auto product = [&](const double* __restrict__ Ablk, const d4 (&Bf)[4], d4 (&acc)[4]) {
    const double* ap = Ablk + lr * TLD + lq;
    double a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = ap[(16 * i) * TLD];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[(ks + 1) & 1][i] = ap[(16 * i) * TLD + 4 * (ks + 1)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks & 1][i], Bf[ks >> 2][ks & 3], acc[i], 0, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    if (t >= nT) continue;            // (uniform; a `break` would keep the loop from unrolling: ring slots in scratch)
    const int p = PT[t], q = QT[t];
    const int n_p = blk_rows(p), ncol = blk_cols(p, q);
    double* B = Bst + (t & 1) * (64 * TLD);
    // the operand block into LDS, rows and columns permuted within their groups of 16
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = fr + 32 * h;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        B[perm4(r) * TLD + perm4(fc + 8 * e)] = (r < n_p && fc + 8 * e < ncol) ? bv[t % kDepth][8 * h + e] : 0.0;
    }
    if (t + kDepth < 10 && t + kDepth < nT)
      fetch(PT[t + kDepth < 10 ? t + kDepth : 9], QT[t + kDepth < 10 ? t + kDepth : 9], bv[t % kDepth]);
    if (q == 0) {
      // G = A_p^T (masked: rows / columns beyond the block are exact zeros); the next panel's rows are requested
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[i][r] = (row_ok && 16 * i + 4 * lq + r < n_p) ? ac[i][r] : 0.0;
      if (p + 1 < np) aload(p + 1 < 4 ? p + 1 : 3, ac);
    }
    __syncthreads();
    CSTAMP(33 + 2 * t);
    if (q < p) {
      product(B, Yn[q < 3 ? q : 0], g);              // G += L_pq (-Y_q)
    } else {
      d4 y[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = (d4){0.0, 0.0, 0.0, 0.0};
      product(B, g, y);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * i + 4 * lq + r;
          if (col >= n_p) y[i][r] = 0.0;                 // padding columns: exact zeros
          if (row_ok && col < n_p) X[(int64_t)myrow * ld + p * pw + col] = y[i][r];
        }
        if (p < 3) Yn[p < 3 ? p : 0][i] = -y[i];
      }
    }
    CSTAMP(34 + 2 * t);
  }
}

// ---------------------------------------------------------------------------
// The update kernel.  One workgroup (WM x WN waves) owns one T x T tile of one
// unit; each wave owns a (T/WM)x(T/WN) block = FMM x FMN MFMA fragments.  K is streamed in steps of 16 through LDS ([row][k], row stride 18
// doubles -> conflict-free ds_read_b64 for the MFMA operand pattern) with the
// next step's global loads issued before the current step's MFMAs.
// ---------------------------------------------------------------------------
template <int T, int BK, int WM, int WN, bool NARROW>
__device__ __forceinline__ void update_body(const UpdTile tl, const UpdUnit& u,
                                            const int64_t* __restrict__ bc_off,
                                            const int* __restrict__ bc_w,
                                            double* __restrict__ L,
                                            const int* __restrict__ relpos,
                                            const int* __restrict__ rlist,
                                            const double* __restrict__ dinv) {
  // LDS row stride BK + 3 doubles (odd): measured best (scripts/update_bench.hip with
  // -DUPD_LDK_PAD=n).  BK + 2 makes the MFMA operand reads conflict-free but the staging
  // writes collide (SQ_LDS_BANK_CONFLICT = 40 % of the LDS-active cycles); the odd
  // strides trade a few read conflicts for conflict-free writes: 32-tile +12 % at
  // K = 1024, 64-tile +1 %, whole factorization -0.7 %.
#ifndef UPD_LDK_PAD
#define UPD_LDK_PAD 3
#endif
  constexpr int LDK = BK + UPD_LDK_PAD;
  constexpr int NT = 64 * WM * WN;    // threads: WM x WN waves
  constexpr int FMM = T / WM / 16;    // MFMA fragments per wave, rows
  constexpr int FMN = T / WN / 16;    // MFMA fragments per wave, columns
  constexpr int PER = T * BK / NT;    // doubles staged per thread per operand per step
  constexpr int TPR = BK / PER;       // threads per tile row
#ifndef UPD_STAGES
#define UPD_STAGES 1
#endif
  // UPD_STAGES LDS stages of both operand tiles (dynamic LDS).  Two stages (step k+1 written into
  // the other stage after the MFMAs of step k: one barrier per step) were measured SLOWER: the
  // 64-tile loses a workgroup per CU to the LDS (58.1 vs 61.3 TFLOP/s at K = 1024, 48.0 vs 51.6
  // at K = 256, scripts/update_bench.hip -DUPD_STAGES=2)
  extern __shared__ __attribute__((aligned(16))) double upd_smem[];
  double* const As = upd_smem;
  double* const Bs = upd_smem + UPD_STAGES * T * LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i0 = tl.ti * T, j0 = tl.tj * T;
  const int M = u.M, N = u.N;
  // A tile with few columns (the remainder of a unit's N beyond its last full tile column: the
  // inter-node updates of the bench workload have N = 132..142 in most units, i.e. 4..14 columns
  // in the third 64-wide tile column) does not need all FMM x FMN fragments per wave:
  //   narrow 1 (<= 32 columns): wave (wm, wn) keeps its rows and takes column fragment wn alone
  //   narrow 2 (<= 16 columns): wave w takes row fragment w and column fragment 0
  // -- half / a quarter of the MFMAs of the step (13 % fewer executed flops in the 64-tile
  // scatter launches of the bench workload).  The 64-tile instance only (k_update below).
  // NARROW is a template parameter: the full tiles run the code they always ran.
  const int narrow = !NARROW ? 0 : (N - j0 <= 16 ? 2 : 1);
  const int na = narrow == 2 ? 1 : FMM, nbf = narrow ? 1 : FMN;   // fragments this wave computes
  // fragment (a, b) of this wave: first row / column inside the tile
  auto roff = [&](int a) { return narrow == 2 ? wave * 16 : wm * (T / WM) + a * 16; };
  auto coff = [&](int b) { return narrow == 2 ? 0 : (narrow == 1 ? wn * 16 : wn * (T / WN) + b * 16); };

  const int srow = tid / TPR;            // tile row staged by this thread
  const int skof = (tid % TPR) * PER;    // first k of its chunk
  const bool rowA_ok = (i0 + srow) < M;
  const bool rowB_ok = (j0 + srow) < N;
  const int rowA = rowA_ok ? i0 + srow : M - 1;   // clamped: always a valid row
  const int rowB = rowB_ok ? j0 + srow : N - 1;

  d4 acc[FMM][FMN];
#pragma unroll
  for (int a = 0; a < FMM; ++a)
#pragma unroll
    for (int b = 0; b < FMN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  // ---- K-segment state -----------------------------------------------------
  int seg = 0, kk = 0, klen = 0;
  const double* aptr = nullptr;
  const double* bptr = nullptr;
  int64_t ldb = 0;
  // width / offset of the NEXT K segment, requested one segment ahead so that a
  // segment switch does not stall on two dependent table loads
  int nxt_w = 0;
  int64_t nxt_off = 0;
  auto seg_setup = [&](int sg) {
    const int bcol = u.src_bcol0 + sg;
    // segment 0 comes with the unit: one dependent lookup less before the first loads
    const int w = sg == 0 ? u.a_w : nxt_w;
    const int64_t base = sg == 0 ? u.a_off : nxt_off;
    if (sg + 1 < u.nseg) {
      nxt_w = bc_w[bcol + 1];
      nxt_off = bc_off[bcol + 1];
    }
    const int rshift = u.seg_r0 + sg * u.seg_stride;
    const int kbeg = (u.nseg == 1) ? u.k0 : 0;
    klen = (u.nseg == 1 && u.klen >= 0) ? u.klen : w;
    aptr = L + base + (int64_t)(u.src_r0 + rowA - rshift) * w + kbeg + skof;
    if (u.mode == MODE_TRSM) {
      ldb = u.dinv_ld;
      bptr = dinv + u.dinv_off + (int64_t)rowB * ldb + skof;
    } else if (u.b_bcol0 >= 0) {
      const int bb = u.b_bcol0 + sg;
      ldb = bc_w[bb];
      bptr = L + bc_off[bb] +
             (int64_t)(u.src_c0 + rowB - (u.b_seg_r0 + sg * u.seg_stride)) * ldb + kbeg + skof;
    } else {
      ldb = w;
      bptr = L + base + (int64_t)(u.src_c0 + rowB - rshift) * w + kbeg + skof;
    }
  };
  double ra[PER], rb[PER];
  // Loads are unconditional (no per-element branches): rows beyond the tile
  // edge re-read the last valid row and K beyond the window re-reads its last
  // column; the values are zeroed afterwards.
  auto load_regs = [&]() {
    const int kleft = klen - (kk + skof);  // valid elements in this thread's chunk
    if (kleft >= PER) {
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        ra[e] = aptr[kk + e];
        rb[e] = bptr[kk + e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < PER; ++e) {
        const int ke = kk + (e < kleft ? e : (kleft > 0 ? kleft - 1 : -skof));
        ra[e] = aptr[ke];
        rb[e] = bptr[ke];
        if (e >= kleft) { ra[e] = 0.0; rb[e] = 0.0; }
      }
    }
    if (!rowA_ok) {
#pragma unroll
      for (int e = 0; e < PER; ++e) ra[e] = 0.0;
    }
    if (!rowB_ok) {
#pragma unroll
      for (int e = 0; e < PER; ++e) rb[e] = 0.0;
    }
  };

  seg_setup(0);
  while (klen <= 0 && seg + 1 < u.nseg) seg_setup(++seg);
  bool more = klen > 0;
  if (more) load_regs();

  // advance to the next K step (possibly the next K segment); false: none left
  auto advance = [&]() {
    kk += BK;
    if (kk < klen) return true;
    while (seg + 1 < u.nseg) {
      seg_setup(++seg);
      kk = 0;
      if (klen > 0) return true;
    }
    return false;
  };
  auto stage = [&](int st) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      As[st * (T * LDK) + srow * LDK + skof + e] = ra[e];
      Bs[st * (T * LDK) + srow * LDK + skof + e] = rb[e];
    }
  };
  // MFMAs of the step in stage st; the operand fragments of sub-step ks+1 are read from LDS
  // before the MFMAs of sub-step ks are issued
  auto mfma_step = [&](int st) {
    const double* Ap = As + st * (T * LDK) + (wm * (T / WM) + (lane & 15)) * LDK + (lane >> 4);
    const double* Bp = Bs + st * (T * LDK) + (wn * (T / WN) + (lane & 15)) * LDK + (lane >> 4);
    // (second fragment set only where it is free: the 128-tile would drop from 4 to 3 waves per
    // SIMD for the 12 extra registers and lose 10 %)
#ifdef UPD_PIPE_ALL
    constexpr bool PIPE = true;
#else
    constexpr bool PIPE = FMM * FMN <= 4;
#endif
    double af[PIPE ? 2 : 1][FMM], bf[PIPE ? 2 : 1][FMN];
    if (PIPE) {
#pragma unroll
      for (int a = 0; a < FMM; ++a) af[0][a] = Ap[a * 16 * LDK];
#pragma unroll
      for (int b = 0; b < FMN; ++b) bf[0][b] = Bp[b * 16 * LDK];
    }
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int cur = PIPE ? (ks & 1) : 0, nx = PIPE ? ((ks + 1) & 1) : 0;
      if (!PIPE || ks + 1 < BK / 4) {
        const int kr = PIPE ? ks + 1 : ks;
#pragma unroll
        for (int a = 0; a < FMM; ++a) af[nx][a] = Ap[a * 16 * LDK + kr * 4];
#pragma unroll
        for (int b = 0; b < FMN; ++b) bf[nx][b] = Bp[b * 16 * LDK + kr * 4];
      }
#pragma unroll
      for (int a = 0; a < FMM; ++a)
#pragma unroll
        for (int b = 0; b < FMN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
    }
  };
  // the narrow tiles: one column fragment per wave, na row fragments
  auto mfma_small = [&](int st) {
    const double* Ap = As + st * (T * LDK) + (roff(0) + (lane & 15)) * LDK + (lane >> 4);
    const double* Bp = Bs + st * (T * LDK) + (coff(0) + (lane & 15)) * LDK + (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const double bv = Bp[ks * 4];
      const double a0 = Ap[ks * 4];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][0], 0, 0, 0);
      if (FMM > 1 && na > 1) {
        const double a1 = Ap[16 * LDK + ks * 4];
        acc[FMM > 1 ? 1 : 0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[FMM > 1 ? 1 : 0][0], 0, 0, 0);
      }
    }
  };
#if UPD_STAGES == 2
  if (more) {
    stage(0);
    more = advance();
    __syncthreads();
    int st = 0;
    while (true) {
      // stage st holds step k; the loads of step k+1 fly during its MFMAs
      const bool nxt = more;
      if (nxt) load_regs();
      if (NARROW) mfma_small(st); else mfma_step(st);
      if (!nxt) break;
      stage(st ^ 1);        // (everybody left this stage before the last barrier)
      more = advance();
      __syncthreads();
      st ^= 1;
    }
  }
#else
  while (more) {
    __syncthreads();
    stage(0);
    __syncthreads();
    more = advance();       // the next K step: its global loads fly during the MFMAs
    if (more) load_regs();
    if (NARROW) mfma_small(0); else mfma_step(0);
  }
#endif

  // ---- epilogue --------------------------------------------------------------
  const int lr = lane >> 4, lc = lane & 15;
  if (u.mode == MODE_SCATTER) {
    // fused expand_buffer: dest[(relpos[i]-r0)*ld + (gcol[j]-c0)] -= acc
    double* D = L + u.d_off;
    int dcol[FMN];
#pragma unroll
    for (int b = 0; b < FMN; ++b) {
      const int j = j0 + coff(b) + lc;
      dcol[b] = (b < nbf && j < N) ? rlist[u.gcol_off + j] - u.d_col0 : -1;
    }
#if defined(SCATTER_EXPERIMENT) && SCATTER_EXPERIMENT == 5
    // (timing only: plain read-modify-write with the loads of a fragment row batched -- what an
    // epilogue that OWNS its destination entries would issue; races between units ignored)
#pragma unroll
    for (int a = 0; a < FMM; ++a) {
      double cv5[4][FMN];
      int64_t dr5[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + roff(a) + lr + 4 * r;
        dr5[r] = (int64_t)(relpos[u.relrow_off + min(i, M - 1)] - u.d_row0) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) cv5[r][b] = D[dr5[r] + (dcol[b] >= 0 ? dcol[b] : 0)];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + roff(a) + lr + 4 * r;
        if (i >= M || a >= na) continue;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + coff(b) + lc;
          if (dcol[b] >= 0 && (!u.lower || u.src_r0 + i >= u.src_c0 + j)) D[dr5[r] + dcol[b]] = cv5[r][b] - acc[a][b][r];
        }
      }
    }
#else
#pragma unroll
    for (int a = 0; a < FMM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + roff(a) + lr + 4 * r;
        if (i >= M || a >= na) continue;
#if defined(SCATTER_EXPERIMENT) && SCATTER_EXPERIMENT == 4
        // (timing only: the 64 lanes of one atomic instruction in ONE destination row, 64 columns
        // side by side, instead of 4 rows x 16 columns -- what a transposed epilogue would issue)
        const int64_t drow = (int64_t)(relpos[u.relrow_off + (i - lr)] - u.d_row0) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + coff(b) + lc;
          if (dcol[b] >= 0 && (!u.lower || u.src_r0 + i >= u.src_c0 + j))
            unsafeAtomicAdd(D + drow + (dcol[b] + 16 * lr) % u.d_ld, -acc[a][b][r]);
        }
#else
        const int64_t drow = (int64_t)(relpos[u.relrow_off + i] - u.d_row0) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + coff(b) + lc;
          if (dcol[b] >= 0 && (!u.lower || u.src_r0 + i >= u.src_c0 + j))
            SCATTER_ADD(D + drow + dcol[b], -acc[a][b][r]);
        }
#endif
      }
#endif
  } else {
    double* D = L + u.d_off + (int64_t)u.d_row0 * u.d_ld + u.d_col0;
    // TRSM writes X in place, BUFFER stores the product into the scratch block: plain stores
    const bool trsm = (u.mode == MODE_TRSM) || (u.mode == MODE_BUFFER);
    const bool atomic = u.atomic != 0;
#pragma unroll
    for (int a = 0; a < FMM; ++a) {
      // The tile owns its destination entries unless the unit says otherwise, so
      // the update is a plain read-modify-write; the 4 * FMN loads of a fragment
      // row are issued together (loads interleaved with the stores would
      // serialise into one global round trip each).  Atomics cost more: the chip
      // adds ~1.3 TB/s of atomic bytes, a third of what plain traffic gets.
      bool ok[4][FMN];
      double cv[4][FMN];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + roff(a) + lr + 4 * r;
        const double* drow = D + (int64_t)min(i, M - 1) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + coff(b) + lc;
          ok[r][b] = a < na && b < nbf && i < M && j < N && (trsm || !u.lower || u.src_r0 + i >= u.src_c0 + j);
          cv[r][b] = (ok[r][b] && !trsm && !atomic) ? drow[j] : 0.0;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + roff(a) + lr + 4 * r;
        double* drow = D + (int64_t)min(i, M - 1) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + coff(b) + lc;
          if (!ok[r][b]) continue;
          if (trsm) drow[j] = acc[a][b][r];
          else if (atomic) unsafeAtomicAdd(drow + j, -acc[a][b][r]);
          else drow[j] = cv[r][b] - acc[a][b][r];
        }
      }
    }
  }
}

template <int T, int BK, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (T == 64 && UPD64_OCC > 0) ? UPD64_OCC : 2) void k_update(const UpdTile* __restrict__ tiles,
                                                const UpdUnit* __restrict__ units,
                                                const int64_t* __restrict__ bc_off,
                                                const int* __restrict__ bc_w,
                                                double* __restrict__ L,
                                                const int* __restrict__ relpos,
                                                const int* __restrict__ rlist,
                                                const double* __restrict__ dinv, int prio) {
  // latency-critical launches (panel chain) outrank the trailing-update waves
  // they share a SIMD with
  if (prio) __builtin_amdgcn_s_setprio(3);
  const UpdTile tl = tiles[blockIdx.x];
  const UpdUnit u = units[tl.unit];
  if (T == 64 && WM == 2 && WN == 2 && u.N - tl.tj * T <= 32)
    update_body<T, BK, WM, WN, (T == 64 && WM == 2 && WN == 2)>(tl, u, bc_off, bc_w, L, relpos, rlist, dinv);
  else
    update_body<T, BK, WM, WN, false>(tl, u, bc_off, bc_w, L, relpos, rlist, dinv);
}

template <int T, int WM, int WN>
__device__ __forceinline__ void update_dma_body(const UpdTile* __restrict__ tiles,
                                                const UpdUnit* __restrict__ units,
                                                const int64_t* __restrict__ bc_off,
                                                const int* __restrict__ bc_w,
                                                double* __restrict__ L,
                                                const int* __restrict__ relpos,
                                                const int* __restrict__ rlist,
                                                const double* __restrict__ dinv, int prio) {
  // latency-critical launches (panel chain) outrank the trailing-update waves
  // they share a SIMD with
  if (prio) __builtin_amdgcn_s_setprio(3);
  constexpr int BK = 16;
  constexpr int NT = 64 * WM * WN;    // threads: WM x WN waves, 16 tile rows per wave
  static_assert(T == 16 * WM * WN, "one wave stages 16 rows of each operand tile per step");
  constexpr int FMM = T / WM / 16;    // MFMA fragments per wave, rows
  constexpr int FMN = T / WN / 16;    // MFMA fragments per wave, columns
  constexpr int ROWB = BK * 8;        // bytes of a tile row in LDS: 128, no padding
  // two stages of [A tile | B tile], rows of 128 bytes whose eight 16-byte chunks are stored
  // XOR-swizzled by (row >> 1) & 7: the DMA writes whole 1 KB runs, the MFMA operand reads
  // (16 rows x one k) still hit 32 different bank pairs
  extern __shared__ __attribute__((aligned(16))) double upd_smem[];
  char* const smem = reinterpret_cast<char*>(upd_smem);

  const UpdTile tl = tiles[blockIdx.x];
  const UpdUnit u = units[tl.unit];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i0 = tl.ti * T, j0 = tl.tj * T;
  const int M = u.M, N = u.N;

  // DMA mapping: instruction i (0, 1) of wave w fills tile rows (2w + i) * 8 + (lane >> 3), LDS
  // chunk lane & 7 of that row, which holds the row's global chunk (lane & 7) ^ ((row >> 1) & 7).
  // Rows beyond the tile edge re-read the last valid row (their results are never stored).
  // (recomputed at every segment switch rather than kept in registers)
  auto dma_row = [&](int i) { return (2 * wave + i) * 8 + (lane >> 3); };

  d4 acc[FMM][FMN];
#pragma unroll
  for (int a = 0; a < FMM; ++a)
#pragma unroll
    for (int b = 0; b < FMN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  // ---- K-segment state -----------------------------------------------------
  int seg = 0, kk = 0, klen = 0;
  const double* aptr[2] = {nullptr, nullptr};
  const double* bptr[2] = {nullptr, nullptr};
  // width / offset of the NEXT K segment, requested one segment ahead so that a
  // segment switch does not stall on two dependent table loads
  int nxt_w = 0;
  int64_t nxt_off = 0;
  auto seg_setup = [&](int sg) {
    const int bcol = u.src_bcol0 + sg;
    // segment 0 comes with the unit: one dependent lookup less before the first loads
    const int w = sg == 0 ? u.a_w : nxt_w;
    const int64_t base = sg == 0 ? u.a_off : nxt_off;
    if (sg + 1 < u.nseg) {
      nxt_w = bc_w[bcol + 1];
      nxt_off = bc_off[bcol + 1];
    }
    const int rshift = u.seg_r0 + sg * u.seg_stride;
    const int kbeg = (u.nseg == 1) ? u.k0 : 0;
    klen = (u.nseg == 1 && u.klen >= 0) ? u.klen : w;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = dma_row(i);
      const int rA = min(i0 + row, M - 1), rB = min(j0 + row, N - 1);
      const int cgl = 2 * ((lane & 7) ^ ((row >> 1) & 7));     // swizzled chunk, in doubles
      aptr[i] = L + base + (int64_t)(u.src_r0 + rA - rshift) * w + kbeg + cgl;
      if (u.mode == MODE_TRSM) {
        bptr[i] = dinv + u.dinv_off + (int64_t)rB * u.dinv_ld + cgl;
      } else if (u.b_bcol0 >= 0) {
        const int bb = u.b_bcol0 + sg;
        const int64_t ldb = bc_w[bb];
        bptr[i] = L + bc_off[bb] + (int64_t)(u.src_c0 + rB - (u.b_seg_r0 + sg * u.seg_stride)) * ldb + kbeg + cgl;
      } else {
        bptr[i] = L + base + (int64_t)(u.src_c0 + rB - rshift) * w + kbeg + cgl;
      }
    }
  };
  // one K step (16 columns from kk of the current segment) of both tiles straight into stage st;
  // a ragged last chunk of a segment is loaded whole (what lies behind the window is some other
  // part of the arena: the allocations carry slack) and cleared in LDS afterwards (clear_tail)
  auto issue = [&](int st) {
    char* base = smem + st * (2 * T * ROWB);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(aptr[i] + kk),
                                       (__attribute__((address_space(3))) void*)(base + (2 * wave + i) * 8 * ROWB),
                                       16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bptr[i] + kk),
                                       (__attribute__((address_space(3))) void*)(base + T * ROWB + (2 * wave + i) * 8 * ROWB),
                                       16, 0, 0);
    }
  };
  auto clear_tail = [&](int st, int rem) {     // columns >= rem of the step in stage st become zero
    char* base = smem + st * (2 * T * ROWB);
    for (int e = tid; e < 2 * T * BK; e += NT) {
      const int row = e >> 4, kc = e & 15;
      if (kc >= rem)
        *(double*)(base + row * ROWB + (((kc >> 1) ^ (((row & (T - 1)) >> 1) & 7)) << 4) + ((kc & 1) << 3)) = 0.0;
    }
  };
  // advance to the next K step (possibly the next K segment); false: none left
  auto advance = [&]() {
    kk += BK;
    if (kk < klen) return true;
    while (seg + 1 < u.nseg) {
      seg_setup(++seg);
      kk = 0;
      if (klen > 0) return true;
    }
    return false;
  };
  auto mfma_step = [&](int st) {
    const char* As = smem + st * (2 * T * ROWB);
    const char* Bs = As + T * ROWB;
    const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double af[FMM], bf[FMN];
      const int kq = 4 * ks + lq, cg = kq >> 1, half = kq & 1;
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
  };

  seg_setup(0);
  while (klen <= 0 && seg + 1 < u.nseg) seg_setup(++seg);
  bool more = klen > 0;
  if (more) {
    issue(0);
    int rem = klen - kk;                       // valid columns of the step in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (rem < BK) { clear_tail(0, rem); __syncthreads(); }
    int st = 0;
    while (true) {
      // stage st holds step k; step k+1 is requested into the other stage before the MFMAs
      const bool nxt = advance();
      if (nxt) {
        issue(st ^ 1);
        rem = klen - kk;
      }
      mfma_step(st);
      if (!nxt) break;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                          // step k+1 has landed; everybody has left stage st
      if (rem < BK) { clear_tail(st ^ 1, rem); __syncthreads(); }
      st ^= 1;
    }
  }

  // ---- epilogue --------------------------------------------------------------
  const int lr = lane >> 4, lc = lane & 15;
  if (u.mode == MODE_SCATTER) {
    // fused expand_buffer: dest[(relpos[i]-r0)*ld + (gcol[j]-c0)] -= acc
    double* D = L + u.d_off;
    int dcol[FMN];
#pragma unroll
    for (int b = 0; b < FMN; ++b) {
      const int j = j0 + wn * (T / WN) + b * 16 + lc;
      dcol[b] = (j < N) ? rlist[u.gcol_off + j] - u.d_col0 : -1;
    }
#pragma unroll
    for (int a = 0; a < FMM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wm * (T / WM) + a * 16 + lr + 4 * r;
        if (i >= M) continue;
#if defined(SCATTER_EXPERIMENT) && SCATTER_EXPERIMENT == 4
        const int64_t drow = (int64_t)(relpos[u.relrow_off + (i - lr)] - u.d_row0) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + wn * (T / WN) + b * 16 + lc;
          if (dcol[b] >= 0 && (!u.lower || u.src_r0 + i >= u.src_c0 + j))
            unsafeAtomicAdd(D + drow + (dcol[b] + 16 * lr) % u.d_ld, -acc[a][b][r]);
        }
#else
        const int64_t drow = (int64_t)(relpos[u.relrow_off + i] - u.d_row0) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + wn * (T / WN) + b * 16 + lc;
          if (dcol[b] >= 0 && (!u.lower || u.src_r0 + i >= u.src_c0 + j))
            SCATTER_ADD(D + drow + dcol[b], -acc[a][b][r]);
        }
#endif
      }
  } else {
    double* D = L + u.d_off + (int64_t)u.d_row0 * u.d_ld + u.d_col0;
    // TRSM writes X in place, BUFFER stores the product into the scratch block: plain stores
    const bool trsm = (u.mode == MODE_TRSM) || (u.mode == MODE_BUFFER);
    const bool atomic = u.atomic != 0;
#pragma unroll
    for (int a = 0; a < FMM; ++a) {
      // The tile owns its destination entries unless the unit says otherwise, so
      // the update is a plain read-modify-write; the 4 * FMN loads of a fragment
      // row are issued together (loads interleaved with the stores would
      // serialise into one global round trip each).  Atomics cost more: the chip
      // adds ~1.3 TB/s of atomic bytes, a third of what plain traffic gets.
      bool ok[4][FMN];
      double cv[4][FMN];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wm * (T / WM) + a * 16 + lr + 4 * r;
        const double* drow = D + (int64_t)min(i, M - 1) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + wn * (T / WN) + b * 16 + lc;
          ok[r][b] = i < M && j < N && (trsm || !u.lower || u.src_r0 + i >= u.src_c0 + j);
          cv[r][b] = (ok[r][b] && !trsm && !atomic) ? drow[j] : 0.0;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wm * (T / WM) + a * 16 + lr + 4 * r;
        double* drow = D + (int64_t)min(i, M - 1) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + wn * (T / WN) + b * 16 + lc;
          if (!ok[r][b]) continue;
          if (trsm) drow[j] = acc[a][b][r];
          else if (atomic) unsafeAtomicAdd(drow + j, -acc[a][b][r]);
          else drow[j] = cv[r][b] - acc[a][b][r];
        }
      }
    }
  }
}


// the 128-tile instantiation, with the register budget (second launch bound = waves per SIMD) that
// keeps 2 workgroups per CU resident.  (A 64-tile instantiation lost to the register-staged 64-tile
// at every K -- 46.9 vs 51.8 TFLOP/s at K = 256 -- and was removed in round 4.)
__global__ __launch_bounds__(512, 4) void k_update_dma128(
    const UpdTile* __restrict__ tiles, const UpdUnit* __restrict__ units, const int64_t* __restrict__ bc_off,
    const int* __restrict__ bc_w, double* __restrict__ L, const int* __restrict__ relpos,
    const int* __restrict__ rlist, const double* __restrict__ dinv, int prio) {
  update_dma_body<128, 4, 2>(tiles, units, bc_off, bc_w, L, relpos, rlist, dinv, prio);
}

// ---------------------------------------------------------------------------
// a26: extend-add of a generated element window into an ancestor tile,
// dest[pos_r(i)][pos_c(j)] -= src[i][j]; positions found by binary search in
// the destination's (sorted) index lists.
// ---------------------------------------------------------------------------
__device__ inline int lower_bound_dev(const int* a, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_scatter_block(int s_m, int s_n, const int* rsrc_index,
                                                       const int* csrc_index, const double* src,
                                                       int lds, const int* rdest_index, int d_m,
                                                       const int* cdest_index, int d_n,
                                                       double* dest, int ldd) {
  const int64_t total = (int64_t)s_m * s_n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int sr = (int)(e / s_n), sc = (int)(e - (int64_t)sr * s_n);
    const int dr = lower_bound_dev(rdest_index, d_m, rsrc_index[sr]);
    const int dc = lower_bound_dev(cdest_index, d_n, csrc_index[sc]);
    dest[(int64_t)dr * ldd + dc] -= src[(int64_t)sr * lds + sc];
  }
}

// ---------------------------------------------------------------------------
// a18 stand-alone: a[row_list[j]*blkn + col_list[i]] += buffer[j*cls + i],
// i < (j < ndiag ? j+1 : cls).  One wavefront per buffer row: the buffer row is
// read coalesced and consecutive col_list entries of one destination row are
// near-contiguous.  Twin of
// reference src/StarPU/expand_buffer_kernels.cu:27-45.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand_buffer(double* __restrict__ a, int blkn,
                                                       const int* __restrict__ row_list, int rls,
                                                       const int* __restrict__ col_list, int cls,
                                                       int ndiag, const double* __restrict__ buffer) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  for (int j = wave; j < rls; j += nwave) {
    const int imax = j < ndiag ? j + 1 : cls;
    double* arow = a + (int64_t)row_list[j] * blkn;
    const double* b = buffer + (int64_t)j * cls;
    for (int i = lane; i < imax; i += 64) arow[col_list[i]] += b[i];
  }
}

// ---------------------------------------------------------------------------
// Triangular solves with the device-resident factor (reference solve_fwd /
// solve_bwd, src/spllt_solve_mod.F90:244-411; per-block kernels
// src/spllt_solve_kernels_mod.F90:11-210).  y is the right-hand side in pivot
// order, overwritten by the solution.  HBM-bound: every entry of L is read once
// per sweep, row-major rows are read by 16 or 64 consecutive lanes.
// ---------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// 16 lanes per row: partial dot products of row[0..len) with NR vectors
// x[q * XS + 0..len) for this lane's residues (k = sub, sub+16, ...), sixteen
// independent loads in flight; every loaded entry of L serves all NR right-hand sides
template <int NR, int XS>
__device__ inline void dot16(const double* __restrict__ row, const double* x, int len, int sub,
                             double (&out)[NR]) {
#pragma unroll
  for (int q = 0; q < NR; ++q) out[q] = 0.0;
  for (int k0 = 0; k0 < len; k0 += 256) {
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = row[min(k0 + sub + 16 * e, len - 1)];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int k = k0 + sub + 16 * e;
      const double a = k < len ? v[e] : 0.0;
      const int kc = min(k, len - 1);
#pragma unroll
      for (int q = 0; q < NR; ++q) out[q] = __builtin_fma(a, x[q * XS + kc], out[q]);
    }
  }
}
__device__ inline double sum16(double v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}

constexpr int kXS = 1024;  // LDS stride between right-hand sides (block column width <= 1024)

// Solve with the diagonal tile of one block column per workgroup, using the
// inverted 64x64 diagonal panels:  forward  x_p = inv(L_pp) (y_p - L_p,<p x_<p),
// backward x_p = inv(L_pp)^T (y_p - L_>p,p^T x_>p).  NR right-hand sides at once
// (y[q * ldy + i]).
template <bool BWD, int NR>
__global__ __launch_bounds__(256) void k_solve_diag(const int* __restrict__ list,
                                                    const SolveUnit* __restrict__ units,
                                                    const double* __restrict__ L,
                                                    const double* __restrict__ dinv,
                                                    const int* __restrict__ rlist,
                                                    double* __restrict__ y, int64_t ldy, const SolveUnit u0, int single) {
  __shared__ double xb[NR * kXS];
  __shared__ double tb[NR * 64];
  __shared__ double part[4][NR * 64];
  // (single: the launch has ONE block column -- every step of the upper levels -- and its descriptor
  // came with the kernel arguments instead of through two dependent loads)
  const SolveUnit u = single ? u0 : units[list[blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 15, rr = tid >> 4;   // 16 lanes per row, 16 rows per pass
  const int w = u.w, pw = u.pw;
  const double* A = L + u.off;
  for (int j = tid; j < w; j += 256) {
    const int gi = u.gcol0 + j;          // (the block column's own columns are consecutive pivot positions)
#pragma unroll
    for (int q = 0; q < NR; ++q) xb[q * kXS + j] = y[q * ldy + gi];
  }
  __syncthreads();
  const int np = (w + pw - 1) / pw;
  for (int pp = 0; pp < np; ++pp) {
    const int p = BWD ? np - 1 - pp : pp;
    const int c0 = p * pw, pn = min(pw, w - c0);
    // inv(L_pp) inside the inverse of its chain block (schedule.hpp winv_offset / winv_ld): the
    // chain block's cw x cw matrix, rows from c0 - g0, columns from c0 - g0
    const int g0 = (c0 / u.cb) * u.cb, ldw = min(u.cb, w - g0);
    int64_t slot = u.dinv_off;
    for (int t = 0; t < g0; t += u.cb) {
      const int64_t cwt = min(u.cb, w - t);
      slot += cwt * cwt;
    }
    const double* D = dinv + slot + (int64_t)(c0 - g0) * ldw + (c0 - g0);
    if (!BWD) {
      // rows of inv(L_pp) for the second half, requested before the first half's loads
      double dv[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dv[r][e] = D[(int64_t)min(rr + 16 * r, pn - 1) * ldw + min(sub + 16 * e, pn - 1)];
      // t_j = y_j - sum_{k<c0} L[c0+j][k] x_k
      if (c0 > 0) {
        double acc[4][NR];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dot16<NR, kXS>(A + (int64_t)(c0 + min(rr + 16 * r, pn - 1)) * w, xb, c0, sub, acc[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const double sacc = sum16(acc[r][q]);
            const int j = rr + 16 * r;
            if (sub == 0 && j < pn) tb[q * 64 + j] = xb[q * kXS + c0 + j] - sacc;
          }
      } else if (tid < pn) {
#pragma unroll
        for (int q = 0; q < NR; ++q) tb[q * 64 + tid] = xb[q * kXS + tid];
      }
      __syncthreads();
      // x_j = sum_{k<=j} Dinv[j][k] t_k   (Dinv is lower triangular, zeros above)
      {
        double acc[4][NR];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            double sa = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = sub + 16 * e;
              sa = __builtin_fma(k < pn ? dv[r][e] : 0.0, tb[q * 64 + min(k, pn - 1)], sa);
            }
            acc[r][q] = sa;
          }
        __syncthreads();   // every read of tb is done before xb/tb change
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            const double sacc = sum16(acc[r][q]);
            const int j = rr + 16 * r;
            if (sub == 0 && j < pn) xb[q * kXS + c0 + j] = sacc;
          }
      }
      __syncthreads();
    } else {
      // t_j = y_j - sum_{k>=c0+pn} L[k][c0+j] x_k : lane = column j, waves split k
      {
        double sa[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) sa[q] = 0.0;
        const int kbeg = c0 + pn, cj = c0 + min(lane, pn - 1);
        for (int k0 = kbeg + wave; k0 < w; k0 += 32) {
          double v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = A[(int64_t)min(k0 + 4 * e, w - 1) * w + cj];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = k0 + 4 * e;
            const double a = k < w ? v[e] : 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) sa[q] = __builtin_fma(a, xb[q * kXS + min(k, w - 1)], sa[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) part[wave][q * 64 + lane] = sa[q];
      }
      __syncthreads();
      if (tid < pn) {
#pragma unroll
        for (int q = 0; q < NR; ++q)
          tb[q * 64 + tid] = xb[q * kXS + c0 + tid] - (part[0][q * 64 + tid] + part[1][q * 64 + tid] +
                                                       part[2][q * 64 + tid] + part[3][q * 64 + tid]);
      }
      __syncthreads();
      // x_j = sum_{k>=j} Dinv[k][j] t_k
      {
        double sa[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) sa[q] = 0.0;
        const int cj = min(lane, pn - 1);
        for (int k0 = wave; k0 < pn; k0 += 32) {
          double v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = D[(int64_t)min(k0 + 4 * e, pn - 1) * ldw + cj];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = k0 + 4 * e;
            const double a = k < pn ? v[e] : 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) sa[q] = __builtin_fma(a, tb[q * 64 + min(k, pn - 1)], sa[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) part[wave][q * 64 + lane] = sa[q];
      }
      __syncthreads();
      if (tid < pn) {
#pragma unroll
        for (int q = 0; q < NR; ++q)
          xb[q * kXS + c0 + tid] = part[0][q * 64 + tid] + part[1][q * 64 + tid] +
                                   part[2][q * 64 + tid] + part[3][q * 64 + tid];
      }
      __syncthreads();
    }
  }
  for (int j = tid; j < w; j += 256) {
    const int gi = u.gcol0 + j;          // (the block column's own columns are consecutive pivot positions)
#pragma unroll
    for (int q = 0; q < NR; ++q) y[q * ldy + gi] = xb[q * kXS + j];
  }
}

// The same for block columns of at most four 64-wide panels (pw = cb = 64, w <= 256: the bench
// configuration's nb = 256) with ONE round trip to memory: the panel steps are a dependent sequence,
// but what they read of L does not depend on them -- every thread requests its share of the whole
// strictly lower part of the diagonal block (96 values) when the kernel starts, and the inverse of the
// next panel while it works on the current one, so a step is LDS reads, FMAs, a 16-lane reduction and
// barriers.  (The general kernel above pays a global round trip per panel step: 24.6 us per 256-wide
// block column forward, 14.1 backward, on the bench workload; 238 dependent launches per solve.)
template <bool BWD, int NR>
__global__ __launch_bounds__(256) void k_solve_diag4(const int* __restrict__ list,
                                                     const SolveUnit* __restrict__ units,
                                                     const double* __restrict__ L,
                                                     const double* __restrict__ dinv,
                                                     const int* __restrict__ rlist,
                                                     double* __restrict__ y, int64_t ldy, const SolveUnit u0, int single) {
  __shared__ double xb[NR * 256];
  __shared__ double tb[NR * 64];
  __shared__ double part[4][NR * 64];
  const SolveUnit u = single ? u0 : units[list[blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 15, rr = tid >> 4;   // forward: 16 lanes per row, 16 rows per pass
  const int w = u.w;
  const int np = (w + 63) >> 6;
  const double* A = L + u.off;
  // ---- everything of L the steps will read ------------------------------------------------
  double lv[3][4][12];       // forward: panel p + 1, row rr + 16 r, columns sub + 16 e (e < 4 (p + 1))
  double bv[3][48];          // backward: panel p, column lane, rows 64 (p + 1) + wave + 4 i (i < 16 (3 - p))
  if (!BWD) {
#pragma unroll
    for (int p = 1; p < 4; ++p) {
      if (p >= np) break;
      const int c0 = 64 * p, pn = min(64, w - c0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* row = A + (int64_t)(c0 + min(rr + 16 * r, pn - 1)) * w;
#pragma unroll
        for (int e = 0; e < 4 * p; ++e) lv[p - 1][r][e] = row[sub + 16 * e];
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (p + 1 >= np) break;
      const int cj = 64 * p + lane;               // (a full panel: p + 1 < np)
#pragma unroll
      for (int i = 0; i < 16 * (3 - p); ++i) {
        const int k = 64 * (p + 1) + wave + 4 * i;
        bv[p][i] = A[(int64_t)min(k, w - 1) * w + cj];
      }
    }
  }
  // inverse of a panel: slot of panel p = dinv_off + sum of the squares of the panels before it (all 64 wide)
  auto wload = [&](int p, double (&dv)[16]) {
    const int pn = min(64, w - 64 * p);
    const double* D = dinv + u.dinv_off + (int64_t)p * 4096;
    if (!BWD) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[4 * r + e] = D[(int64_t)min(rr + 16 * r, pn - 1) * pn + min(sub + 16 * e, pn - 1)];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) dv[e] = D[(int64_t)min(wave + 4 * e, pn - 1) * pn + min(lane, pn - 1)];
    }
  };
  double dv[16], dvn[16];
  wload(BWD ? np - 1 : 0, dv);
  for (int j = tid; j < w; j += 256) {
    const int gi = u.gcol0 + j;          // (the block column's own columns are consecutive pivot positions)
#pragma unroll
    for (int q = 0; q < NR; ++q) xb[q * 256 + j] = y[q * ldy + gi];
  }
  __syncthreads();
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    if (pp >= np) break;
    const int p = BWD ? np - 1 - pp : pp;
    const int c0 = 64 * p, pn = min(64, w - c0);
    if (pp + 1 < np) wload(BWD ? p - 1 : p + 1, dvn);
    if (!BWD) {
      // t_j = y_j - sum_{k < c0} L[c0 + j][k] x_k
      if (pp > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            double sa = 0.0;
#pragma unroll
            for (int e = 0; e < 4 * pp; ++e) sa = __builtin_fma(lv[pp - 1][r][e], xb[q * 256 + sub + 16 * e], sa);
            sa = sum16(sa);
            const int j = rr + 16 * r;
            if (sub == 0 && j < pn) tb[q * 64 + j] = xb[q * 256 + c0 + j] - sa;
          }
      } else if (tid < pn) {
#pragma unroll
        for (int q = 0; q < NR; ++q) tb[q * 64 + tid] = xb[q * 256 + tid];
      }
      __syncthreads();
      // x_j = sum_{k <= j} Dinv[j][k] t_k
      double acc[4][NR];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          double sa = 0.0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = sub + 16 * e;
            sa = __builtin_fma(k < pn ? dv[4 * r + e] : 0.0, tb[q * 64 + min(k, pn - 1)], sa);
          }
          acc[r][q] = sum16(sa);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = rr + 16 * r;
        if (sub == 0 && j < pn) {
#pragma unroll
          for (int q = 0; q < NR; ++q) xb[q * 256 + c0 + j] = acc[r][q];
        }
      }
      __syncthreads();
    } else {
      // t_j = y_j - sum_{k >= c0 + pn} L[k][c0 + j] x_k : lane = column j, the waves split k
      if (pp > 0) {
        double sa[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) sa[q] = 0.0;
        // (p = np - 1 - pp: the rows below are those of the pp panels behind it; bv[p] was loaded for
        // i < 16 (3 - p), of which the first 16 pp exist)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          if (pc != p) continue;
#pragma unroll
          for (int i = 0; i < 16 * (3 - pc); ++i) {
            const int k = 64 * (pc + 1) + wave + 4 * i;
            const double a = k < w ? bv[pc][i] : 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) sa[q] = __builtin_fma(a, xb[q * 256 + min(k, w - 1)], sa[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) part[wave][q * 64 + lane] = sa[q];
        __syncthreads();
        if (tid < pn) {
#pragma unroll
          for (int q = 0; q < NR; ++q)
            tb[q * 64 + tid] = xb[q * 256 + c0 + tid] - (part[0][q * 64 + tid] + part[1][q * 64 + tid] +
                                                         part[2][q * 64 + tid] + part[3][q * 64 + tid]);
        }
      } else if (tid < pn) {
#pragma unroll
        for (int q = 0; q < NR; ++q) tb[q * 64 + tid] = xb[q * 256 + c0 + tid];
      }
      __syncthreads();
      // x_j = sum_{k >= j} Dinv[k][j] t_k
      {
        double sa[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) sa[q] = 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k = wave + 4 * e;
          const double a = k < pn ? dv[e] : 0.0;
#pragma unroll
          for (int q = 0; q < NR; ++q) sa[q] = __builtin_fma(a, tb[q * 64 + min(k, pn - 1)], sa[q]);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) part[wave][q * 64 + lane] = sa[q];
      }
      __syncthreads();
      if (tid < pn) {
#pragma unroll
        for (int q = 0; q < NR; ++q)
          xb[q * 256 + c0 + tid] = part[0][q * 64 + tid] + part[1][q * 64 + tid] +
                                   part[2][q * 64 + tid] + part[3][q * 64 + tid];
      }
      __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) dv[e] = dvn[e];
  }
  for (int j = tid; j < w; j += 256) {
    const int gi = u.gcol0 + j;          // (the block column's own columns are consecutive pivot positions)
#pragma unroll
    for (int q = 0; q < NR; ++q) y[q * ldy + gi] = xb[q * 256 + j];
  }
}

// Rows below the diagonal tile, one strip of kSolveStripRows (64) rows per workgroup.
//   forward : y[idx[r]] -= sum_k L[r][k] x_k      (x = solved entries of this block column)
//   backward: y[idx[k]] -= sum_r L[r][k] x[idx[r]]
template <bool BWD, int NR>
__global__ __launch_bounds__(256) void k_solve_strip(const UpdTile* __restrict__ tiles,
                                                     const SolveUnit* __restrict__ units,
                                                     const double* __restrict__ L,
                                                     const int* __restrict__ rlist,
                                                     double* __restrict__ y, int64_t ldy, const SolveUnit u0, int single) {
  __shared__ double xb[NR * kXS];
  // (single: all strips of the launch belong to ONE block column, strip i = workgroup i)
  const int ti = single ? (int)blockIdx.x : (int)tiles[blockIdx.x].ti;
  const SolveUnit u = single ? u0 : units[tiles[blockIdx.x].unit];
  const int tid = threadIdx.x;
  const int w = u.w;
  const int r0 = w + ti * kSolveStripRows;
  const int nr = min(kSolveStripRows, u.nrow - r0);
  const double* A = L + u.off + (int64_t)r0 * w;
  const int* idx = rlist + u.idx_off;
  if (!BWD) {
    for (int k = tid; k < w; k += 256) {
      const int gi = u.gcol0 + k;
#pragma unroll
      for (int q = 0; q < NR; ++q) xb[q * kXS + k] = y[q * ldy + gi];
    }
    __syncthreads();
    // 16 lanes per row (coalesced 128-byte reads), 4 rows per thread in flight
    const int sub = tid & 15, rr = tid >> 4;
    double acc[4][NR];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      dot16<NR, kXS>(A + (int64_t)min(rr + 16 * r, nr - 1) * w, xb, w, sub, acc[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rr + 16 * r;
      const int gi = idx[r0 + min(row, nr - 1)];
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const double sacc = sum16(acc[r][q]);
        if (sub == 0 && row < nr) unsafeAtomicAdd(y + q * ldy + gi, -sacc);
      }
    }
  } else {
    for (int r = tid; r < nr; r += 256) {
      const int gi = idx[r0 + r];
#pragma unroll
      for (int q = 0; q < NR; ++q) xb[q * kXS + r] = y[q * ldy + gi];
    }
    __syncthreads();
    for (int k = tid; k < w; k += 256) {
      double sa[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) sa[q] = 0.0;
      for (int q0 = 0; q0 < nr; q0 += 8) {
        double v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = A[(int64_t)min(q0 + e, nr - 1) * w + k];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const double a = q0 + e < nr ? v[e] : 0.0;
#pragma unroll
          for (int q = 0; q < NR; ++q) sa[q] = __builtin_fma(a, xb[q * kXS + min(q0 + e, nr - 1)], sa[q]);
        }
      }
      const int gi = u.gcol0 + k;
#pragma unroll
      for (int q = 0; q < NR; ++q) unsafeAtomicAdd(y + q * ldy + gi, -sa[q]);
    }
  }
}

template <int NR>
static void launch_solve_nr(hipStream_t st, int kind, const int* list, const UpdTile* tiles,
                            int64_t first, int64_t count, const SolveUnit* units, const double* L,
                            const double* dinv, const int* rlist, double* y, int64_t ldy, bool four,
                            const SolveUnit* one) {
  const dim3 g((unsigned)count), b(256);
  const SolveUnit u0 = one ? *one : SolveUnit{};
  const int single = one ? 1 : 0;
  switch (kind) {
    case SV_DIAG_FWD:
      if (four)
        hipLaunchKernelGGL((k_solve_diag4<false, NR>), g, b, 0, st, list + first, units, L, dinv, rlist, y, ldy, u0, single);
      else
        hipLaunchKernelGGL((k_solve_diag<false, NR>), g, b, 0, st, list + first, units, L, dinv, rlist, y, ldy, u0, single);
      break;
    case SV_DIAG_BWD:
      if (four)
        hipLaunchKernelGGL((k_solve_diag4<true, NR>), g, b, 0, st, list + first, units, L, dinv, rlist, y, ldy, u0, single);
      else
        hipLaunchKernelGGL((k_solve_diag<true, NR>), g, b, 0, st, list + first, units, L, dinv, rlist, y, ldy, u0, single);
      break;
    case SV_STRIP_FWD:
      hipLaunchKernelGGL((k_solve_strip<false, NR>), g, b, 0, st, tiles + first, units, L, rlist, y, ldy, u0, single);
      break;
    default:
      hipLaunchKernelGGL((k_solve_strip<true, NR>), g, b, 0, st, tiles + first, units, L, rlist, y, ldy, u0, single);
      break;
  }
}

// nr = 1, 2 or 4 right-hand sides per sweep: y[q * ldy + i]
void launch_solve(hipStream_t st, int kind, const int* list, const UpdTile* tiles, int64_t first,
                  int64_t count, const SolveUnit* units, const double* L, const double* dinv,
                  const int* rlist, double* y, int nr, int64_t ldy, bool four, const SolveUnit* one) {
  if (count <= 0) return;
  if (nr >= 4)
    launch_solve_nr<4>(st, kind, list, tiles, first, count, units, L, dinv, rlist, y, ldy, four, one);
  else if (nr >= 2)
    launch_solve_nr<2>(st, kind, list, tiles, first, count, units, L, dinv, rlist, y, ldy, four, one);
  else
    launch_solve_nr<1>(st, kind, list, tiles, first, count, units, L, dinv, rlist, y, ldy, four, one);
}

// ---------------------------------------------------------------------------
// launch wrappers (plain C++ callers do not see HIP launch syntax)
// ---------------------------------------------------------------------------
void launch_expand_buffer(hipStream_t st, double* a, int blkn, const int* row_list, int rls,
                          const int* col_list, int cls, int ndiag, const double* buffer) {
  if (rls <= 0 || cls <= 0) return;
  int blocks = (rls + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_expand_buffer, dim3(blocks), dim3(256), 0, st, a, blkn, row_list, rls,
                     col_list, cls, ndiag, buffer);
}

// ---------------------------------------------------------------------------
// a8 once more (spllt_init_node, kernels_mod:2301-2364: clear the block columns, copy A in), as
// ONE pass over the arena: a workgroup builds 32 KB of it in LDS -- zeros, then the entries of A
// that fall into the chunk (the val -> L map bucketed by chunk once per pattern: 6 bytes per entry)
// -- and writes the chunk out with 16-byte stores.  The arena is written once instead of being
// cleared (hipMemsetAsync: 229 us for the 1.57 GB of the bench workload) and then hit by 11 M
// scattered 8-byte stores (k_scatter_val: 221 us).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_arena(double* __restrict__ L, int64_t arena,
                                                    const double* __restrict__ val,
                                                    const int64_t* __restrict__ cptr,
                                                    const unsigned short* __restrict__ loc,
                                                    const int* __restrict__ src) {
  __shared__ __attribute__((aligned(16))) double img[kInitChunk];
  const int tid = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int64_t e0 = cptr[c], e1 = cptr[c + 1];
  for (int e = tid; e < kInitChunk; e += 256) img[e] = 0.0;
  __syncthreads();
  for (int64_t e = e0 + tid; e < e1; e += 256) img[loc[e]] = val[src[e]];
  __syncthreads();
  const int64_t base = c * kInitChunk;
  const int64_t n = arena - base < kInitChunk ? arena - base : kInitChunk;
  typedef double d2v __attribute__((ext_vector_type(2)));
  if (n == kInitChunk) {
    d2v* out = reinterpret_cast<d2v*>(L + base);
    const d2v* in = reinterpret_cast<const d2v*>(img);
    for (int e = tid; e < kInitChunk / 2; e += 256) out[e] = in[e];
  } else {
    for (int64_t e = tid; e < n; e += 256) L[base + e] = img[e];
  }
}

void launch_init_arena(const LaunchSink& st, double* L, int64_t arena, const double* val, const int64_t* cptr,
                       const unsigned short* loc, const int* src) {
  if (arena <= 0) return;
  const int64_t chunks = (arena + kInitChunk - 1) / kInitChunk;
  emit(st, k_init_arena, dim3((unsigned)chunks), dim3(256), 0, L, arena, val, cptr, loc, src);
}

void launch_scatter_val(const LaunchSink& st, double* L, const double* val, const int64_t* dst,
                        const int64_t* src, int64_t n) {
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  emit(st, k_scatter_val, dim3((unsigned)blocks), dim3(256), 0, L, val, dst, src, n);
}

void launch_potrf(hipStream_t st, const PotrfUnit* units, int64_t count, double* L, double* dinv,
                  int* flag, const PotrfUnit& unit0) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_potrf_panel, dim3((unsigned)count), dim3(256), 0, st, units, L, dinv, flag, unit0);
}

void launch_chain_panel(const LaunchSink& st, const ChainUnit* units, int64_t count, double* L, double* dinv,
                        int* flag, const ChainUnit& unit0) {
  if (count <= 0) return;
  emit(st, k_chain_potrf, dim3((unsigned)count), dim3(256), 0, units, L, dinv, flag, unit0);
}

void launch_chain_block(const LaunchSink& st, const ChainUnit* units, int64_t count, double* L, double* dinv,
                        int* flag, int pw, const ChainUnit& unit0) {
  if (count <= 0) return;
  const unsigned lds = (unsigned)(sizeof(PotrfShared) + sizeof(double) * 2 * 64 * TLD);
  thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    (void)hipFuncSetAttribute((const void*)k_chain_block, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_dev = dev;
  }
  emit(st, k_chain_block, dim3((unsigned)count), dim3(kPanelThreads), lds, units, L, dinv, flag, pw, unit0);
}

void launch_subtree(const LaunchSink& st, const SubTask* tasks, int64_t count, const SubNode* nodes,
                    const UpdUnit* units, const int* relpos, const int* rlist, double* L, double* dinv, double* gen,
                    int* flag) {
  if (count <= 0) return;
  emit(st, k_subtree, dim3((unsigned)count), dim3(kSubThreads), 0, tasks, nodes, units, relpos, rlist, L, dinv, gen, flag);
}

void launch_trsm_rows(const LaunchSink& st, const UpdTile* tiles, int64_t count, const UpdUnit* units, double* L,
                      const double* dinv, int pw, int prio) {
  if (count <= 0) return;
  const unsigned lds = (unsigned)(sizeof(double) * 2 * 64 * TLD);
  thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    (void)hipFuncSetAttribute((const void*)k_trsm_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_dev = dev;
  }
  emit(st, k_trsm_rows, dim3((unsigned)count), dim3(256), lds, tiles, units, L, dinv, pw, prio);
}

void launch_panel(const LaunchSink& st, const UpdTile* tiles, int64_t count, const PanelUnit* units, double* L,
                  double* dinv, int* counters, int* flag) {
  if (count <= 0) return;
  const unsigned lds = (unsigned)(sizeof(PotrfShared) + sizeof(double) * 2 * 64 * TLD);
  thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    (void)hipFuncSetAttribute((const void*)k_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_dev = dev;
  }
  emit(st, k_panel, dim3((unsigned)count), dim3(kPanelThreads), lds, tiles, units, L, dinv, counters, flag);
}

// ---------------------------------------------------------------------------
// Deterministic assembly: dest tile -= sum over its items (in list order) of the buffered
// update blocks.  a18 spllt_expand_buffer (kernels_mod:2010-2053) turned around: the
// reference walks a buffer and adds into the destination (one task at a time per
// destination, task_mod:1239-1241); here one workgroup owns a destination tile and walks
// the buffers that hit it, so no two writers ever meet and the order of the adds is fixed.
// HBM-bound: 8 B per buffered entry + 16 B per destination entry.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const GatherTile* __restrict__ tiles,
                                                const GatherItem* __restrict__ items,
                                                double* __restrict__ L,
                                                const double* __restrict__ scratch,
                                                const int* __restrict__ relpos,
                                                const int* __restrict__ rlist) {
  __shared__ double acc[64 * 65];
  const GatherTile t = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 65; e += 256) acc[e] = 0.0;
  __syncthreads();
  for (int it = 0; it < t.count; ++it) {
    const GatherItem g = items[t.first + it];
    const int ni = g.i1 - g.i0, nj = g.j1 - g.j0;
    const double* buf = scratch + g.buf_off;
    for (int e = tid; e < ni * nj; e += 256) {
      const int i = g.i0 + e / nj, j = g.j0 + e % nj;
      if (g.lower && g.diag_shift + i < j) continue;
      const int r = relpos[g.relrow_off + i] - t.drow_base - t.row0;
      const int c = rlist[g.gcol_off + j] - t.dcol_base - t.col0;
      acc[r * 65 + c] += buf[(int64_t)i * g.ld + j];   // (i, j) -> (r, c) is injective inside an item
    }
    __syncthreads();   // the next item may hit the same entries
  }
  double* D = L + t.d_off + (int64_t)t.row0 * t.d_ld + t.col0;
  for (int e = tid; e < t.rows * t.cols; e += 256) {
    const int r = e / t.cols, c = e % t.cols;
    const double a = acc[r * 65 + c];
    if (a != 0.0) D[(int64_t)r * t.d_ld + c] -= a;
  }
}

void launch_gather(const LaunchSink& st, const GatherTile* tiles, int64_t count, const GatherItem* items,
                   double* L, const double* scratch, const int* relpos, const int* rlist) {
  if (count <= 0) return;
  emit(st, k_gather, dim3((unsigned)count), dim3(256), 0, tiles, items, L, scratch, relpos, rlist);
}

// multi-GPU: the "not positive definite" flag travels with the exchange buffer.  Before the
// exchange every rank appends 1.0 (a pivot of its subtrees failed) or 0.0; after the sum
// over the ranks a non-zero entry marks the factorization as failed on EVERY rank.
__global__ void k_flag_pack(const int* __restrict__ flag, double* __restrict__ slot) {
  *slot = (*flag != 0x7fffffff) ? 1.0 : 0.0;
}
__global__ void k_flag_unpack(const double* __restrict__ slot, int* __restrict__ flag) {
  if (*slot > 0.5 && *flag == 0x7fffffff) *flag = 0x7ffffffe;   // failed on another rank
}
__global__ __launch_bounds__(256) void k_mask(double* __restrict__ y, const double* __restrict__ keep, int n,
                                              int nrhs, int64_t ldy) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double k = keep[i];
  for (int q = 0; q < nrhs; ++q) y[q * ldy + i] *= k;
}
void launch_mask(hipStream_t st, double* y, const double* keep, int n, int nrhs, int64_t ldy) {
  if (n <= 0 || nrhs <= 0) return;
  hipLaunchKernelGGL(k_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, keep, n, nrhs, ldy);
}
void launch_flag_pack(hipStream_t st, const int* flag, double* slot) {
  hipLaunchKernelGGL(k_flag_pack, dim3(1), dim3(1), 0, st, flag, slot);
}
void launch_flag_unpack(hipStream_t st, const double* slot, int* flag) {
  hipLaunchKernelGGL(k_flag_unpack, dim3(1), dim3(1), 0, st, slot, flag);
}

// debug aid (engine flag 128): one workgroup per CU-sized LDS allocation writes a
// signalling-NaN pattern over all 160 KB, so that any kernel that later reads LDS it
// has not written computes with NaNs instead of with whatever the previous kernel left
__global__ __launch_bounds__(256) void k_poison_lds(int* sink) {
  extern __shared__ __attribute__((aligned(16))) double poison_smem[];
  const double snan = __longlong_as_double(0x7FF4DEADBEEF0001LL);
  for (int i = threadIdx.x; i < 160 * 1024 / 8; i += 256) poison_smem[i] = snan;
  __syncthreads();
  if (sink && poison_smem[threadIdx.x] == 0.0) *sink = 1;   // keeps the stores alive
}

void launch_poison_lds(hipStream_t st) {
  thread_local int attr_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != attr_dev) {
    (void)hipFuncSetAttribute((const void*)k_poison_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_dev = dev;
  }
  hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(256), 160 * 1024, st, (int*)nullptr);
}

void launch_update(const LaunchSink& st, int tile, const UpdTile* tiles, int64_t count,
                   const UpdUnit* units, const int64_t* bc_off, const int* bc_w, double* L,
                   const int* relpos, const int* rlist, const double* dinv, int prio,
                   int lds_pad, bool allow_dma, bool latency) {
  if (count <= 0) return;
  static const int lat_bk = [] { const char* e = std::getenv("SPLLT_LAT_BK"); return e ? std::atoi(e) : 0; }();
  if (latency && lat_bk == 64 && (tile == 64 || tile == 32)) {
    thread_local int attr_dev3 = -1;
    int dev3 = 0;
    (void)hipGetDevice(&dev3);
    if (dev3 != attr_dev3) {
      (void)hipFuncSetAttribute((const void*)k_update<64, 64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute((const void*)k_update<32, 64, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_dev3 = dev3;
    }
    const unsigned lds = (unsigned)(sizeof(double) * 2 * UPD_STAGES * tile * (64 + UPD_LDK_PAD));
    if (tile == 64)
      emit(st, k_update<64, 64, 2, 2>, dim3((unsigned)count), dim3(256), lds, tiles, units, bc_off, bc_w, L, relpos,
           rlist, dinv, prio);
    else
      emit(st, k_update<32, 64, 2, 2>, dim3((unsigned)count), dim3(256), lds, tiles, units, bc_off, bc_w, L, relpos,
           rlist, dinv, prio);
    return;
  }
  // lds_pad: extra (unused) dynamic LDS that caps the workgroups per CU of a
  // trailing-update launch so that panel-chain kernels find room beside it
  // dynamic LDS: the operand stages of the tile (+ the optional pad)
  auto lds_of = [&](int T, int BK) {
    return (unsigned)(sizeof(double) * 2 * UPD_STAGES * T * (BK + UPD_LDK_PAD)) + (lds_pad > 0 ? (unsigned)lds_pad : 0u);
  };
  const unsigned pad = tile == 128 ? lds_of(128, UPD128_BK) : tile == 64 ? lds_of(64, 16) : lds_of(32, 32);
  {
    // the padded launches exceed the default 64 KB of LDS per workgroup; the
    // attribute is per device, so it is (re)applied for the current one
    thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != attr_dev) {
      (void)hipFuncSetAttribute((const void*)k_update<128, UPD128_BK, UPD128_WM, UPD128_WN>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute((const void*)k_update<64, 16, 2, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute((const void*)k_update<32, 32, 2, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_dev = dev;
    }
  }
  // operand tiles straight into LDS (k_update_dma128) for the 128-tiles; SPLLT_UPD_DMA=0: the
  // register-staged kernel
  static const bool use_dma = [] {
    const char* e = std::getenv("SPLLT_UPD_DMA");
    return !(e && std::atoi(e) == 0);
  }();
  if (use_dma && allow_dma && tile == 128) {
    const unsigned lds = (unsigned)(2 * 2 * tile * 128) + (lds_pad > 0 ? (unsigned)lds_pad : 0u);
    thread_local int attr_dev2 = -1;
    int dev2 = 0;
    (void)hipGetDevice(&dev2);
    if (dev2 != attr_dev2) {
      (void)hipFuncSetAttribute((const void*)k_update_dma128, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_dev2 = dev2;
    }
    emit(st, k_update_dma128, dim3((unsigned)count), dim3(512), lds, tiles, units, bc_off, bc_w, L, relpos, rlist,
         dinv, prio);
    return;
  }
  if (tile == 128)
    emit(st, k_update<128, UPD128_BK, UPD128_WM, UPD128_WN>, dim3((unsigned)count), dim3(64 * UPD128_WM * UPD128_WN),
         pad, tiles, units, bc_off, bc_w, L, relpos, rlist, dinv, prio);
  else if (tile == 64)
    emit(st, k_update<64, 16, 2, 2>, dim3((unsigned)count), dim3(256), pad, tiles, units, bc_off, bc_w, L, relpos,
         rlist, dinv, prio);
  else
    emit(st, k_update<32, 32, 2, 2>, dim3((unsigned)count), dim3(256), pad, tiles, units, bc_off, bc_w, L, relpos,
         rlist, dinv, prio);
}

void launch_scatter_block(hipStream_t st, int s_m, int s_n, const int* rsrc_index,
                          const int* csrc_index, const double* src, int lds,
                          const int* rdest_index, int d_m, const int* cdest_index, int d_n,
                          double* dest, int ldd) {
  int64_t total = (int64_t)s_m * s_n;
  if (total <= 0) return;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scatter_block, dim3((unsigned)blocks), dim3(256), 0, st, s_m, s_n,
                     rsrc_index, csrc_index, src, lds, rdest_index, d_m, cdest_index, d_n, dest,
                     ldd);
}

}  // namespace spx
