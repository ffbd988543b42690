// Hand-written CDNA4 (gfx950) kernels of the factorize hot path.
//
//   k_scatter_val   a8  spllt_init_node   (reference src/spllt_kernels_mod.F90:2301-2364)
//   k_potrf_panel   a11 spllt_factor_diag_block (:1168-1189) on <=64-wide panels,
//                   also emits the inverse of the factored panel for the TRSM
//   k_update<T>     a12 spllt_solve_block (:1217) as X = A * inv(L_pp)^T,
//                   a13 spllt_update_block (:1261-1292),
//                   a16+a18 spllt_update_between + spllt_expand_buffer
//                   (:2108-2237, :2010-2053) with the scatter fused into the
//                   GEMM epilogue -- one fp64-MFMA kernel, three epilogues.
//   k_scatter_block a26 spllt_scatter_block (:1122-1160) extend-add
//   k_solve_diag / k_solve_strip   forward / backward substitution on the device-resident
//                   factor (reference src/spllt_solve_mod.F90), up to 4 right-hand sides
//   experimental variants of the panel chain (not in the default program, kept with
//   their tests because the measurements in DESIGN.md refer to them):
//   k_trsm_strip (flag 4), k_tile_chain (flag 4), k_panel_step (flag 32)
//
// Storage convention: every block column of L is a
// row-major (rows x width) matrix; all products are C = A * B^T with both
// operands K-contiguous, which is exactly the operand order
// v_mfma_f64_16x16x4_f64 wants: lane l supplies A[l&15][l>>4] and B[l>>4][l&15]
// and receives C[(l>>4) + 4r][l&15] in register r.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace spx {

typedef double d4 __attribute__((ext_vector_type(4)));

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
#else
#define STAMP(i) do { } while (0)
#endif

struct PotrfShared {
  double T[64 * TLD];
  double X[64 * TLD];
  double DI[4][16 * DLD];
  double RI[64];  // reciprocals of the diagonal of L
};

// Factor (and invert) one <=64 x <=64 block held at A (row stride ld); the whole
// workgroup takes part.  D receives inv(L) (row-major, ld = n).
__device__ __forceinline__ void potrf64_body(PotrfShared& sh, double* __restrict__ A, int ld, int n,
                                             double* __restrict__ D, int gcol, int flags,
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
  // identity-padded lower triangle: thread t owns 16 consecutive columns of row t/4
  const int li = tid >> 2, lj0 = (tid & 3) * 16;
  {
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = lj0 + e;
      v[e] = (li < n && j <= li) ? A[(int64_t)li * ld + j] : ((li == j) ? 1.0 : 0.0);
    }
    if (li < np) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        T[li * TLD + lj0 + e] = v[e];
        X[li * TLD + lj0 + e] = 0.0;
        if (!do_chol && lj0 + e == li) RI[li] = 1.0 / v[e];
      }
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
        for (int k = 0; k < J * 16; k += 4) {
          const double a = -T[(I * 16 + lr) * TLD + k + lq];
          const double b = T[(J * 16 + lr) * TLD + k + lq];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
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
        for (int k = j + 1; k < 16; ++k) tk[k] = bcast(row[j], k);
        double djj = bcast(row[j], j);
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
        if (failcol != (1 << 30) && lane == 0) atomicMin(flag, u.gcol + J * 16 + failcol + 1);
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
  if (li < n) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = lj0 + e;
      if (j < n) {
        if (do_chol && j <= li) A[(int64_t)li * ld + j] = T[li * TLD + j];
        D[li * n + j] = X[li * TLD + j];
      }
    }
  }
  STAMP(16);
}

__global__ __launch_bounds__(256) void k_potrf_panel(const PotrfUnit* __restrict__ units,
                                                     double* __restrict__ L,
                                                     double* __restrict__ dinv,
                                                     int* __restrict__ flag) {
  __shared__ PotrfShared sh;
  __builtin_amdgcn_s_setprio(3);
  const PotrfUnit u = units[blockIdx.x];
  potrf64_body(sh, L + u.off, u.ld, u.n, dinv + u.dinv_off, u.gcol, u.flags, flag);
}

// ---------------------------------------------------------------------------
// The whole panel chain of one diagonal tile (w <= 256) in ONE workgroup:
//   for every 64-wide panel p:  update the panel's block column inside the tile
//   by the previous panels, factor + invert its diagonal block, solve the
//   tile rows below it.
// It replaces 3*np-1 dependent launches on the critical path (each of which has
// to win CU slots against the concurrently running trailing update) by a single
// resident workgroup.  The tile-local products read their MFMA operands
// straight from global memory (the tile is 512 KB and L2-resident).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_chain(const PotrfUnit* __restrict__ units,
                                                    double* __restrict__ L,
                                                    double* __restrict__ dinv,
                                                    int* __restrict__ flag) {
  __shared__ PotrfShared sh;
  const PotrfUnit u = units[blockIdx.x];   // off = block column, n = tile order, flags = panel width
  const int w = u.ld, nt = u.n, pw = u.flags;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
  double* A = L + u.off;
  int64_t slot = u.dinv_off;
  for (int c0 = 0; c0 < nt; c0 += pw) {
    const int pn = min(pw, nt - c0);
    const int nct = (pn + 15) >> 4;
    if (c0 > 0) {
      // (a) A[r][c0+j] -= sum_{k<c0} A[r][k] A[c0+j][k],  r in [c0, nt), lower part
      const int nrt = (nt - c0 + 15) >> 4;
      for (int t = wave; t < nrt * nct; t += 4) {
        const int rt = t / nct, ct = t - rt * nct;
        if (rt < ct) continue;  // entirely above the diagonal
        const int rbase = c0 + rt * 16, cbase = c0 + ct * 16;
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + lq + 4 * r, col = cbase + lr;
          acc[r] = (row < nt && col < c0 + pn) ? A[(int64_t)row * w + col] : 0.0;
        }
        const int ra = min(rbase + lr, nt - 1), rb = min(cbase + lr, nt - 1);
        const double* pa = A + (int64_t)ra * w + lq;
        const double* pb = A + (int64_t)rb * w + lq;
        // c0 is a multiple of the panel width (64): 16 k-steps per chunk, all 32
        // operand loads of a chunk in flight before its MFMAs
        for (int k0 = 0; k0 < c0; k0 += 64) {
          double av[16], bv[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int k = k0 + 4 * t;
            av[t] = k < c0 ? pa[k] : 0.0;
            bv[t] = k < c0 ? pb[k] : 0.0;
          }
#pragma unroll
          for (int t = 0; t < 16; ++t)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[t], bv[t], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + lq + 4 * r, col = cbase + lr;
          if (row < nt && col < c0 + pn && row >= col) A[(int64_t)row * w + col] = acc[r];
        }
      }
      __syncthreads();
    }
    // (b) factor + invert the diagonal block of the panel
    potrf64_body(sh, A + (int64_t)c0 * w + c0, w, pn, dinv + slot, u.gcol + c0, 0, flag);
    __syncthreads();
    // (c) rows below the panel inside the tile: X = A * inv(L_pp)^T, one 16-row
    // tile per wave (its K operand is loaded completely before it is overwritten)
    {
      const double* D = dinv + slot;
      const int r1 = c0 + pn;
      const int nrt = (nt - r1 + 15) >> 4;
      const int ksteps = (pn + 3) >> 2;
      for (int rt = wave; rt < nrt; rt += 4) {
        const int rbase = r1 + rt * 16;
        const int ra = min(rbase + lr, nt - 1);
        double av[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int k = 4 * t + lq;
          const double v = A[(int64_t)ra * w + c0 + min(k, pn - 1)];
          av[t] = (t < ksteps && k < pn && rbase + lr < nt) ? v : 0.0;
        }
        for (int ct = 0; ct < nct; ++ct) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
          const int jrow = min(ct * 16 + lr, pn - 1);
          double bv[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int k = min(4 * t + lq, pn - 1);
            bv[t] = D[jrow * pn + k];
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int k = 4 * t + lq;
            const double b = (t < ksteps && k < pn && ct * 16 + lr < pn) ? bv[t] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], b, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rbase + lq + 4 * r, col = ct * 16 + lr;
            if (row < nt && col < pn) A[(int64_t)row * w + c0 + col] = acc[r];
          }
        }
      }
    }
    __syncthreads();
    slot += (int64_t)pn * pn;
  }
}

// ---------------------------------------------------------------------------
// One panel step below its POTRF (PanelStepUnit).  Workgroup = 32 rows:
//   Xi = A_i * inv(L_pp)^T                     (the TRSM of its rows, stored)
//   Xd = A_d * inv(L_pp)^T                     (rows of the next panel's diagonal
//                                               block, recomputed by every workgroup)
//   D[i, next panel] -= [S_i | O_i | Xi] * [S_d | O_d | Xd]^T
// with S = previous block column, O = source block column's panels before p
// (MFMA operands straight from global/L2) and the panel itself from LDS.  It
// replaces the TRSM launch, the left-looking update launch of the next panel
// and the separate block-column c -> c+1 update on the critical path.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_panel_step(const UpdTile* __restrict__ tiles,
                                                    const PanelStepUnit* __restrict__ units,
                                                    double* __restrict__ L,
                                                    const double* __restrict__ dinv) {
  __shared__ double Ai[32 * TLD];
  __shared__ double Ad[64 * TLD];
  __builtin_amdgcn_s_setprio(2);
  const UpdTile tl = tiles[blockIdx.x];
  const PanelStepUnit u = units[tl.unit];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int pn = u.pn, ld = u.ld;
  const int rb = u.c0 + pn;          // first stored row below the diagonal block
  const int i0 = tl.ti * 32;         // first row of this tile inside the region
  const bool has_dest = u.d_off >= 0;
  const double* A = L + u.off;
  // ---- stage A_i (32 x pn) and A_d (d_pn x pn), zero padded ----------------
  {
    const int r = tid >> 3, k0 = (tid & 7) * 8;
    const bool ok = i0 + r < u.nrows;
    const double* src = A + (int64_t)(rb + (ok ? i0 + r : 0)) * ld + u.c0;
    double v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[min(k0 + e, pn - 1)];
#pragma unroll
    for (int e = 0; e < 8; ++e) Ai[r * TLD + k0 + e] = (ok && k0 + e < pn) ? v[e] : 0.0;
  }
  if (has_dest) {
    const int r = tid >> 2, k0 = (tid & 3) * 16;
    const bool ok = r < u.d_pn;
    const double* src = A + (int64_t)(rb + (ok ? r : 0)) * ld + u.c0;
    double v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = src[min(k0 + e, pn - 1)];
#pragma unroll
    for (int e = 0; e < 16; ++e) Ad[r * TLD + k0 + e] = (ok && k0 + e < pn) ? v[e] : 0.0;
  }
  __syncthreads();
  // ---- Xi, Xd = A * Dinv^T ---------------------------------------------------
  const double* D = dinv + u.dinv_off;
  const int rf = w & 1, cfb = (w >> 1) * 2;   // Xi / update fragments of this wave: (rf, cfb), (rf, cfb+1)
  d4 xd[4], xi[2];
#pragma unroll
  for (int cf = 0; cf < 4; ++cf) {
    double bv[16];
    const int jrow = min(cf * 16 + lr, pn - 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) bv[t] = D[jrow * pn + min(4 * t + lq, pn - 1)];
    const bool jok = cf * 16 + lr < pn;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    d4 acci = {0.0, 0.0, 0.0, 0.0};
    const bool mine = (cf >> 1) == (w >> 1);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const double b = (jok && 4 * t + lq < pn) ? bv[t] : 0.0;
      if (has_dest) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ad[(w * 16 + lr) * TLD + 4 * t + lq], b, acc, 0, 0, 0);
      if (mine) acci = __builtin_amdgcn_mfma_f64_16x16x4f64(Ai[(rf * 16 + lr) * TLD + 4 * t + lq], b, acci, 0, 0, 0);
    }
    xd[cf] = acc;
    if (mine) xi[cf & 1] = acci;
  }
  __syncthreads();
  // ---- write back: Xd -> Ad, Xi -> Ai and L (final values of the panel rows) --
#pragma unroll
  for (int cf = 0; cf < 4; ++cf)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ad[(w * 16 + lq + 4 * r) * TLD + cf * 16 + lr] = xd[cf][r];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rf * 16 + lq + 4 * r, col = (cfb + c) * 16 + lr;
      Ai[row * TLD + col] = xi[c][r];
      if (i0 + row < u.nrows && col < pn)
        L[u.off + (int64_t)(rb + i0 + row) * ld + u.c0 + col] = xi[c][r];
    }
  if (!has_dest) return;
  __syncthreads();
  // ---- update of the next panel ------------------------------------------------
  d4 acc[2];
  acc[0] = (d4){0.0, 0.0, 0.0, 0.0};
  acc[1] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const double a = Ai[(rf * 16 + lr) * TLD + 4 * t + lq];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ad[((cfb + 0) * 16 + lr) * TLD + 4 * t + lq], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Ad[((cfb + 1) * 16 + lr) * TLD + 4 * t + lq], acc[1], 0, 0, 0);
  }
  const bool aok = i0 + rf * 16 + lr < u.nrows;
  const int arow = rb + (aok ? i0 + rf * 16 + lr : 0);
  const bool b0ok = (cfb + 0) * 16 + lr < u.d_pn, b1ok = (cfb + 1) * 16 + lr < u.d_pn;
  const int b0row = rb + (b0ok ? (cfb + 0) * 16 + lr : 0);
  const int b1row = rb + (b1ok ? (cfb + 1) * 16 + lr : 0);
  for (int sg = 0; sg < 2; ++sg) {
    // sg 0: the source block column's panels before p;  sg 1: the previous block column
    const int K = sg == 0 ? u.c0 : (u.s_off >= 0 ? u.s_k : 0);
    if (K <= 0) continue;
    const int sld = sg == 0 ? ld : u.s_ld;
    const int rsh = sg == 0 ? 0 : u.s_rshift;
    const double* base = L + (sg == 0 ? u.off : u.s_off);
    const double* pa = base + (int64_t)(arow + rsh) * sld + lq;
    const double* p0 = base + (int64_t)(b0row + rsh) * sld + lq;
    const double* p1 = base + (int64_t)(b1row + rsh) * sld + lq;
    for (int k0 = 0; k0 < K; k0 += 64) {
      double av[16], v0[16], v1[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int k = min(k0 + 4 * t, K - 4 + 3 - lq);   // stays inside the row: k + lq <= K - 1
        av[t] = pa[k];
        v0[t] = p0[k];
        v1[t] = p1[k];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool kok = k0 + 4 * t + lq < K;
        const double a = (aok && kok) ? av[t] : 0.0;
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (b0ok && kok) ? v0[t] : 0.0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (b1ok && kok) ? v1[t] : 0.0, acc[1], 0, 0, 0);
      }
    }
  }
  double* Dst = L + u.d_off;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + rf * 16 + lq + 4 * r, j = (cfb + c) * 16 + lr;
      if (i < u.nrows && j < u.d_pn && i >= j)
        unsafeAtomicAdd(Dst + (int64_t)(rb + i - u.d_rshift) * u.d_ld + u.d_c0 + j, -acc[c][r]);
    }
}

// ---------------------------------------------------------------------------
// The update kernel.  One workgroup (WM x WN waves) owns one T x T tile of one
// unit; each wave owns a (T/WM)x(T/WN) block = FMM x FMN MFMA fragments.  K is streamed in steps of 16 through LDS ([row][k], row stride 18
// doubles -> conflict-free ds_read_b64 for the MFMA operand pattern) with the
// next step's global loads issued before the current step's MFMAs.
// ---------------------------------------------------------------------------
template <int T, int BK, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void k_update(const UpdTile* __restrict__ tiles,
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
  __shared__ double As[T * LDK];
  __shared__ double Bs[T * LDK];

  const UpdTile tl = tiles[blockIdx.x];
  const UpdUnit u = units[tl.unit];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int i0 = tl.ti * T, j0 = tl.tj * T;
  const int M = u.M, N = u.N;

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

  while (more) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      As[srow * LDK + skof + e] = ra[e];
      Bs[srow * LDK + skof + e] = rb[e];
    }
    __syncthreads();
    // advance to the next K step and start its global loads
    kk += BK;
    if (kk >= klen) {
      more = false;
      while (seg + 1 < u.nseg) {
        seg_setup(++seg);
        kk = 0;
        if (klen > 0) { more = true; break; }
      }
    }
    if (more) load_regs();
    // MFMAs of the staged step
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      double af[FMM], bf[FMN];
      const int kq = ks * 4 + (lane >> 4);
#pragma unroll
      for (int a = 0; a < FMM; ++a) af[a] = As[(wm * (T / WM) + a * 16 + (lane & 15)) * LDK + kq];
#pragma unroll
      for (int b = 0; b < FMN; ++b) bf[b] = Bs[(wn * (T / WN) + b * 16 + (lane & 15)) * LDK + kq];
#pragma unroll
      for (int a = 0; a < FMM; ++a)
#pragma unroll
        for (int b = 0; b < FMN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
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
        const int64_t drow = (int64_t)(relpos[u.relrow_off + i] - u.d_row0) * u.d_ld;
#pragma unroll
        for (int b = 0; b < FMN; ++b) {
          const int j = j0 + wn * (T / WN) + b * 16 + lc;
          if (dcol[b] >= 0 && (!u.lower || u.src_r0 + i >= u.src_c0 + j))
            unsafeAtomicAdd(D + drow + dcol[b], -acc[a][b][r]);
        }
      }
  } else {
    double* D = L + u.d_off + (int64_t)u.d_row0 * u.d_ld + u.d_col0;
    const bool trsm = (u.mode == MODE_TRSM);
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

// ---------------------------------------------------------------------------
// a12 for all sub-diagonal rows of a block column in one launch: each
// workgroup owns a strip of RS rows and runs the blocked substitution over the
// block column's panels with the strip resident in LDS:
//     X_p = (A_p - sum_{q<p} X_q L_pq^T) * inv(L_pp)^T ,   p = 0, 1, ...
// L_pq (blocks of the factored diagonal tile) and inv(L_pp) (dinv scratch) are
// staged through LDS; all products run on v_mfma_f64_16x16x4_f64, wave w owning
// the 16-column tile w of the current panel.  Replaces the 2*np-1 dependent
// TRSM/UPDATE launches per block column of the unfused path.
// ---------------------------------------------------------------------------
constexpr int SLD = 66;  // staging block row stride

// 64 x 64 staging block: thread t fetches 16 consecutive doubles of row t/4
// with all loads in flight at once (clamped addresses, zero-filled afterwards)
__device__ inline void stage64(double* __restrict__ Ls, const double* __restrict__ src, int ld,
                               int nrow, int ncol, int tid) {
  const int j = tid >> 2, c = (tid & 3) * 16;
  const int jc = j < nrow ? j : nrow - 1;
  double v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int k = c + e < ncol ? c + e : ncol - 1;
    v[e] = src[(int64_t)jc * ld + k];
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) Ls[j * SLD + c + e] = (j < nrow && c + e < ncol) ? v[e] : 0.0;
}

template <int RS, int WMAX>
__global__ __launch_bounds__(256) void k_trsm_strip(const UpdTile* __restrict__ tiles,
                                                    const StripUnit* __restrict__ units,
                                                    double* __restrict__ L,
                                                    const double* __restrict__ dinv) {
  constexpr int XLD = WMAX + 2;
  constexpr int RT = RS / 16;
  constexpr int TPR = 256 / RS;          // threads per strip row
  __shared__ double Xs[RS * XLD];
  __shared__ double Ls[64 * SLD];
  const UpdTile tl = tiles[blockIdx.x];
  const StripUnit u = units[tl.unit];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
  const int w = u.ld, pw = u.pw;
  const int wpad = (w + 15) & ~15;
  const int r0 = u.row0 + (int)tl.ti * RS;
  const int nr = min(RS, u.row0 + u.nrows - r0);
  double* A = L + u.off;
  // strip load: thread t owns row t/TPR, 16-column chunks (t%TPR), (t%TPR)+TPR, ...
  const int si = tid / TPR, sc = (tid % TPR) * 16;
  {
    const double* arow = A + (int64_t)(r0 + (si < nr ? si : nr - 1)) * w;
    for (int c = sc; c < wpad; c += TPR * 16) {
      double v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = arow[c + e < w ? c + e : w - 1];
#pragma unroll
      for (int e = 0; e < 16; ++e) Xs[si * XLD + c + e] = (si < nr && c + e < w) ? v[e] : 0.0;
    }
    // A ragged last panel makes the 16-wide accumulator tiles and the K loop (steps of
    // 4) read up to 63 columns past the block column.  Those operands meet zero
    // factors, but stale LDS may hold NaN bit patterns and 0 * NaN poisons the whole
    // output row (seen once as a spurious "not positive definite"): clear them.
    for (int c = wpad + sc; c < min(wpad + 64, XLD); c += TPR * 16)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (c + e < XLD) Xs[si * XLD + c + e] = 0.0;
  }
  const int np = (w + pw - 1) / pw;
  int64_t slot = u.dinv_off;
  for (int p = 0; p < np; ++p) {
    const int c0 = p * pw;
    const int pn = min(pw, w - c0);
    const bool act = wave * 16 < pn;   // this wave's 16-column tile exists in the panel
    d4 acc[RT];
    __syncthreads();                   // Xs complete (initial load / previous panel's X)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (act) acc[rt] = ld_c_s<XLD>(Xs, rt * 16, c0 + wave * 16, lane);
    for (int q = 0; q < p; ++q) {
      __syncthreads();                 // previous staging block fully consumed
      stage64(Ls, A + (int64_t)c0 * w + q * pw, w, pn, pw, tid);
      __syncthreads();
      if (act) {
        for (int k = 0; k < pw; k += 4) {
          const double b = Ls[(wave * 16 + lr) * SLD + k + lq];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const double a = -Xs[(rt * 16 + lr) * XLD + q * pw + k + lq];
            acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[rt], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    // a panel whose width is not a multiple of 16 ends inside a wave's tile: the
    // columns beyond it belong to the next panel and must not be touched
    const bool colok = wave * 16 + lr < pn;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (act && colok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(rt * 16 + lq + 4 * r) * XLD + c0 + wave * 16 + lr] = acc[rt][r];
      }
    // stage inv(L_pp), zero-padded to 64 x 64
    stage64(Ls, dinv + slot, pn, pn, pn, tid);
    __syncthreads();
    d4 res[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) res[rt] = (d4){0.0, 0.0, 0.0, 0.0};
    if (act) {
      const int kend = (pn + 3) & ~3;
      for (int k = 0; k < kend; k += 4) {
        const double b = Ls[(wave * 16 + lr) * SLD + k + lq];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const double a = Xs[(rt * 16 + lr) * XLD + c0 + k + lq];
          res[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, res[rt], 0, 0, 0);
        }
      }
    }
    __syncthreads();                   // every wave has read the panel before it is overwritten
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      if (act && colok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[(rt * 16 + lq + 4 * r) * XLD + c0 + wave * 16 + lr] = res[rt][r];
      }
    slot += (int64_t)pn * pn;
  }
  __syncthreads();
  if (si < nr) {
    double* arow = A + (int64_t)(r0 + si) * w;
    for (int c = sc; c < w; c += TPR * 16) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (c + e < w) arow[c + e] = Xs[si * XLD + c + e];
    }
  }
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
                                                    double* __restrict__ y, int64_t ldy) {
  __shared__ double xb[NR * kXS];
  __shared__ double tb[NR * 64];
  __shared__ double part[4][NR * 64];
  const SolveUnit u = units[list[blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 15, rr = tid >> 4;   // 16 lanes per row, 16 rows per pass
  const int w = u.w, pw = u.pw;
  const double* A = L + u.off;
  const int* idx = rlist + u.idx_off;
  for (int j = tid; j < w; j += 256) {
    const int gi = idx[j];
#pragma unroll
    for (int q = 0; q < NR; ++q) xb[q * kXS + j] = y[q * ldy + gi];
  }
  __syncthreads();
  const int np = (w + pw - 1) / pw;
  for (int pp = 0; pp < np; ++pp) {
    const int p = BWD ? np - 1 - pp : pp;
    const int c0 = p * pw, pn = min(pw, w - c0);
    int64_t slot = u.dinv_off + (int64_t)p * pw * pw;  // panels before p are full width
    const double* D = dinv + slot;
    if (!BWD) {
      // rows of inv(L_pp) for the second half, requested before the first half's loads
      double dv[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dv[r][e] = D[(int64_t)min(rr + 16 * r, pn - 1) * pn + min(sub + 16 * e, pn - 1)];
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
          for (int e = 0; e < 8; ++e) v[e] = D[min(k0 + 4 * e, pn - 1) * pn + cj];
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
    const int gi = idx[j];
#pragma unroll
    for (int q = 0; q < NR; ++q) y[q * ldy + gi] = xb[q * kXS + j];
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
                                                     double* __restrict__ y, int64_t ldy) {
  __shared__ double xb[NR * kXS];
  const UpdTile tl = tiles[blockIdx.x];
  const SolveUnit u = units[tl.unit];
  const int tid = threadIdx.x;
  const int w = u.w;
  const int r0 = w + (int)tl.ti * kSolveStripRows;
  const int nr = min(kSolveStripRows, u.nrow - r0);
  const double* A = L + u.off + (int64_t)r0 * w;
  const int* idx = rlist + u.idx_off;
  if (!BWD) {
    for (int k = tid; k < w; k += 256) {
      const int gi = idx[k];
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
      const int gi = idx[k];
#pragma unroll
      for (int q = 0; q < NR; ++q) unsafeAtomicAdd(y + q * ldy + gi, -sa[q]);
    }
  }
}

template <int NR>
static void launch_solve_nr(hipStream_t st, int kind, const int* list, const UpdTile* tiles,
                            int64_t first, int64_t count, const SolveUnit* units, const double* L,
                            const double* dinv, const int* rlist, double* y, int64_t ldy) {
  const dim3 g((unsigned)count), b(256);
  switch (kind) {
    case SV_DIAG_FWD:
      hipLaunchKernelGGL((k_solve_diag<false, NR>), g, b, 0, st, list + first, units, L, dinv, rlist, y, ldy);
      break;
    case SV_DIAG_BWD:
      hipLaunchKernelGGL((k_solve_diag<true, NR>), g, b, 0, st, list + first, units, L, dinv, rlist, y, ldy);
      break;
    case SV_STRIP_FWD:
      hipLaunchKernelGGL((k_solve_strip<false, NR>), g, b, 0, st, tiles + first, units, L, rlist, y, ldy);
      break;
    default:
      hipLaunchKernelGGL((k_solve_strip<true, NR>), g, b, 0, st, tiles + first, units, L, rlist, y, ldy);
      break;
  }
}

// nr = 1, 2 or 4 right-hand sides per sweep: y[q * ldy + i]
void launch_solve(hipStream_t st, int kind, const int* list, const UpdTile* tiles, int64_t first,
                  int64_t count, const SolveUnit* units, const double* L, const double* dinv,
                  const int* rlist, double* y, int nr, int64_t ldy) {
  if (count <= 0) return;
  if (nr >= 4)
    launch_solve_nr<4>(st, kind, list, tiles, first, count, units, L, dinv, rlist, y, ldy);
  else if (nr >= 2)
    launch_solve_nr<2>(st, kind, list, tiles, first, count, units, L, dinv, rlist, y, ldy);
  else
    launch_solve_nr<1>(st, kind, list, tiles, first, count, units, L, dinv, rlist, y, ldy);
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

void launch_scatter_val(hipStream_t st, double* L, const double* val, const int64_t* dst,
                        const int64_t* src, int64_t n) {
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scatter_val, dim3((unsigned)blocks), dim3(256), 0, st, L, val, dst, src, n);
}

void launch_potrf(hipStream_t st, const PotrfUnit* units, int64_t count, double* L, double* dinv,
                  int* flag) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_potrf_panel, dim3((unsigned)count), dim3(256), 0, st, units, L, dinv, flag);
}

void launch_tile_chain(hipStream_t st, const PotrfUnit* units, int64_t count, double* L,
                       double* dinv, int* flag) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_tile_chain, dim3((unsigned)count), dim3(256), 0, st, units, L, dinv, flag);
}

void launch_update(hipStream_t st, int tile, const UpdTile* tiles, int64_t count,
                   const UpdUnit* units, const int64_t* bc_off, const int* bc_w, double* L,
                   const int* relpos, const int* rlist, const double* dinv, int prio,
                   int lds_pad) {
  if (count <= 0) return;
  // lds_pad: extra (unused) dynamic LDS that caps the workgroups per CU of a
  // trailing-update launch so that panel-chain kernels find room beside it
  const unsigned pad = lds_pad > 0 ? (unsigned)lds_pad : 0u;
  if (pad > 0) {
    // the padded launches exceed the default 64 KB of LDS per workgroup; the
    // attribute is per device, so it is (re)applied for the current one
    thread_local int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != attr_dev) {
      (void)hipFuncSetAttribute((const void*)k_update<128, 16, 4, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      (void)hipFuncSetAttribute((const void*)k_update<64, 16, 2, 2>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      attr_dev = dev;
    }
  }
  if (tile == 128)
    hipLaunchKernelGGL((k_update<128, 16, 4, 2>), dim3((unsigned)count), dim3(512), pad, st, tiles,
                       units, bc_off, bc_w, L, relpos, rlist, dinv, prio);
  else if (tile == 64)
    hipLaunchKernelGGL((k_update<64, 16, 2, 2>), dim3((unsigned)count), dim3(256), pad, st, tiles,
                       units, bc_off, bc_w, L, relpos, rlist, dinv, prio);
  else
    hipLaunchKernelGGL((k_update<32, 32, 2, 2>), dim3((unsigned)count), dim3(256), pad, st, tiles,
                       units, bc_off, bc_w, L, relpos, rlist, dinv, prio);
}

void launch_panel_step(hipStream_t st, const UpdTile* tiles, int64_t count,
                       const PanelStepUnit* units, double* L, const double* dinv) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_panel_step, dim3((unsigned)count), dim3(256), 0, st, tiles, units, L, dinv);
}

void launch_strip(hipStream_t st, int rs, const UpdTile* tiles, int64_t count,
                  const StripUnit* units, double* L, const double* dinv) {
  if (count <= 0) return;
  if (rs == 32)
    hipLaunchKernelGGL((k_trsm_strip<32, 320>), dim3((unsigned)count), dim3(256), 0, st, tiles, units, L, dinv);
  else
    hipLaunchKernelGGL((k_trsm_strip<16, 896>), dim3((unsigned)count), dim3(256), 0, st, tiles, units, L, dinv);
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
