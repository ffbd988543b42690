// Per-kernel operator API of include/spllt_hip.h, group (1): stream-taking
// twins of the reference's bind(C) factor kernels
// (src/spllt_kernels_mod.F90:1193,1233,1295,2055,2241; CUDA twin
// src/StarPU/expand_buffer_kernels.cu:48-62).  They drive the same gfx950
// kernels as the batched engine through tiny one-off work tables.  Because the
// tables are uploaded per call, these twins synchronise `stream` before they
// return; the asynchronous, batched form of the same kernels is the engine.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>

#include "kernels.hpp"
#include "schedule.hpp"
#include "spllt_hip.h"

using namespace spx;

namespace {

inline int64_t abs_off(const void* p) { return (int64_t)(reinterpret_cast<intptr_t>(p) / 8); }

struct OpsBatch {
  std::vector<int64_t> bc_off;
  std::vector<int> bc_w;
  Program P;

  int run(hipStream_t st, int* dev_flag, const int* relpos, const int* rlist) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
      std::fprintf(stderr, "spllt-hip: no HIP device available\n");
      return SPLLT_ERROR_HIP;
    }
    int64_t* d_off = nullptr;
    int* d_w = nullptr;
    UpdUnit* d_units = nullptr;
    UpdTile* d_tiles = nullptr;
    PotrfUnit* d_potrf = nullptr;
    ChainUnit* d_chain = nullptr;
    PanelUnit* d_panel = nullptr;
    int* d_pcnt = nullptr;
    double* d_dinv = nullptr;
    int* d_flag = dev_flag;
    bool own_flag = false;
    hipError_t e = hipSuccess;
    auto up = [&](void** d, const void* h, size_t bytes) {
      if (e != hipSuccess) return;
      e = hipMalloc(d, std::max<size_t>(bytes, 8));
      if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    for (UpdUnit& u : P.units) {  // segment 0 of every unit carries its block column
      u.a_off = bc_off[(size_t)u.src_bcol0];
      u.a_w = bc_w[(size_t)u.src_bcol0];
    }
    up((void**)&d_off, bc_off.data(), bc_off.size() * sizeof(int64_t));
    up((void**)&d_w, bc_w.data(), bc_w.size() * sizeof(int));
    up((void**)&d_units, P.units.data(), P.units.size() * sizeof(UpdUnit));
    up((void**)&d_tiles, P.tiles.data(), P.tiles.size() * sizeof(UpdTile));
    up((void**)&d_potrf, P.potrf_units.data(), P.potrf_units.size() * sizeof(PotrfUnit));
    up((void**)&d_chain, P.chain_units.data(), P.chain_units.size() * sizeof(ChainUnit));
    up((void**)&d_panel, P.panel_units.data(), P.panel_units.size() * sizeof(PanelUnit));
    {
      std::vector<int> zeros(2 * std::max<size_t>(1, P.panel_units.size()), 0);
      up((void**)&d_pcnt, zeros.data(), zeros.size() * sizeof(int));
    }
    if (e == hipSuccess) e = hipMalloc((void**)&d_dinv, sizeof(double) * (std::max<int64_t>(1, P.dinv_size) + 32));
    // (the POTRF kernels store the lower triangle of an inverse only: the rest of a slot stays zero)
    if (e == hipSuccess) e = hipMemset(d_dinv, 0, sizeof(double) * (std::max<int64_t>(1, P.dinv_size) + 32));
    if (e == hipSuccess && !d_flag) {
      own_flag = true;
      int big = INT_MAX;
      e = hipMalloc((void**)&d_flag, sizeof(int));
      if (e == hipSuccess) e = hipMemcpy(d_flag, &big, sizeof(int), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
      double* base = nullptr;  // table offsets are absolute (address / 8)
      for (const Launch& l : P.launches) {
        if (l.count <= 0) continue;
        if (l.kind == L_POTRF)
          launch_potrf(st, d_potrf + l.first, l.count, base, d_dinv, d_flag, P.potrf_units[(size_t)l.first]);
        else if (l.kind == L_CHAIN)
          launch_chain_panel(st, d_chain + l.first, l.count, base, d_dinv, d_flag, P.chain_units[(size_t)l.first]);
        else if (l.kind == L_CHAIN4)
          launch_chain_block(st, d_chain + l.first, l.count, base, d_dinv, d_flag, P.pw, P.chain_units[(size_t)l.first]);
        else if (l.kind == L_TRSM4)
          launch_trsm_rows(st, d_tiles + l.first, l.count, d_units, base, d_dinv, P.pw, 0);
        else if (l.kind == L_PANEL)
          launch_panel(st, d_tiles + l.first, l.count, d_panel, base, d_dinv, d_pcnt, d_flag);
        else
          launch_update(st, l.tile, d_tiles + l.first, l.count, d_units, d_off, d_w, base, relpos,
                        rlist, d_dinv, 0, 0, false);   // (caller-owned tiles: no slack for whole-chunk loads)
      }
      e = hipGetLastError();
      hipError_t e2 = hipStreamSynchronize(st);
      if (e == hipSuccess) e = e2;
    }
    hipFree(d_off); hipFree(d_w); hipFree(d_units); hipFree(d_tiles); hipFree(d_potrf); hipFree(d_chain); hipFree(d_panel); hipFree(d_pcnt); hipFree(d_dinv);
    if (own_flag) hipFree(d_flag);
    if (e != hipSuccess) {
      std::fprintf(stderr, "spllt-hip: operator failed: %s\n", hipGetErrorString(e));
      return SPLLT_ERROR_HIP;
    }
    return 0;
  }
};

// a single supernode of m rows and n columns stored as ONE block column
void one_bcol_symbolic(Symbolic& S, int m, int n, int64_t off) {
  S.n = n;
  S.nnodes = 1;
  S.nb = std::max(n, 1);
  S.sptr = {0, n};
  S.sparent = {1};
  S.rptr = {0, m};
  S.rlist.resize(m);
  std::iota(S.rlist.begin(), S.rlist.end(), 0);
  S.level = {0};
  S.node_bcol0 = {0, 1};
  BlockCol b{};
  b.node = 0; b.width = n; b.r0 = 0; b.nrow = m; b.off = off; b.blk0 = 0;
  S.bcols = {b};
}

void push_gemm(Program& P, const UpdUnit& u, int T) {
  Launch L{};
  L.kind = L_GEMM;
  L.first = (int64_t)P.tiles.size();
  L.tile = T;
  int uid = (int)P.units.size();
  P.units.push_back(u);
  for (int tj = 0; tj < (u.N + T - 1) / T; ++tj)
    for (int ti = 0; ti < (u.M + T - 1) / T; ++ti) {
      if (u.lower && u.src_r0 + (ti + 1) * T - 1 < u.src_c0 + tj * T) continue;
      UpdTile t;
      t.unit = uid; t.ti = (short)ti; t.tj = (short)tj;
      P.tiles.push_back(t);
    }
  L.count = (int64_t)P.tiles.size() - L.first;
  if (L.count > 0) P.launches.push_back(L);
}

int tile_for(int M, int N) { return (M >= 96 && N >= 96) ? 128 : 64; }

}  // namespace

extern "C" {

int spllt_factor_diag_block_hip(void* stream, int m, int n, double* bc, int* dev_flag) {
  if (m < n || n <= 0 || !bc) return SPLLT_ERROR_PARAMETER;
  Symbolic S;
  one_bcol_symbolic(S, m, n, abs_off(bc));
  OpsBatch B;
  ScheduleOptions so;
  so.lookahead = false;  // one stream, program order
  so.subtrees = false;   // (the batch runner knows the panel-chain launches only)
  build_program(S, so, B.P);
  B.bc_off = {S.bcols[0].off};
  B.bc_w = {n};
  return B.run((hipStream_t)stream, dev_flag, nullptr, nullptr);
}

int spllt_solve_block_hip(void* stream, int m, int n, const double* bc_kk, double* bc_ik) {
  if (m <= 0 || n <= 0 || !bc_kk || !bc_ik) return SPLLT_ERROR_PARAMETER;
  OpsBatch B;
  B.bc_off = {abs_off(bc_ik), abs_off(bc_kk)};
  B.bc_w = {n, n};
  const int pw = kPanelMax;
  // inverses of the diagonal panels of L_kk (no factorization: flags bit 0)
  {
    Launch L{};
    L.kind = L_POTRF;
    L.first = 0;
    int64_t slot = 0;
    for (int c0 = 0; c0 < n; c0 += pw) {
      int pn = std::min(pw, n - c0);
      PotrfUnit q{};
      q.off = abs_off(bc_kk) + (int64_t)c0 * n + c0;
      q.ld = n; q.n = pn; q.gcol = c0; q.flags = 1; q.dinv_off = slot;
      slot += (int64_t)pn * pn;
      B.P.potrf_units.push_back(q);
    }
    B.P.dinv_size = slot;
    L.count = (int64_t)B.P.potrf_units.size();
    B.P.launches.push_back(L);
  }
  int64_t slot = 0;
  for (int c0 = 0; c0 < n; c0 += pw) {
    int pn = std::min(pw, n - c0);
    if (c0 > 0) {  // X_p -= X[:, :c0] * L_kk[c0:c0+pn, :c0]^T
      UpdUnit u{};
      u.mode = MODE_DIRECT; u.lower = 0;
      u.d_off = abs_off(bc_ik); u.d_ld = n; u.d_row0 = 0; u.d_col0 = c0;
      u.src_bcol0 = 0; u.nseg = 1; u.seg_r0 = 0; u.seg_stride = n; u.src_r0 = 0; u.M = m;
      u.b_bcol0 = 1; u.b_seg_r0 = 0; u.src_c0 = c0; u.N = pn;
      u.k0 = 0; u.klen = c0;
      push_gemm(B.P, u, tile_for(u.M, u.N));
    }
    UpdUnit t{};
    t.mode = MODE_TRSM; t.lower = 0; t.b_bcol0 = -1;
    t.d_off = abs_off(bc_ik); t.d_ld = n; t.d_row0 = 0; t.d_col0 = c0;
    t.src_bcol0 = 0; t.nseg = 1; t.seg_r0 = 0; t.seg_stride = n; t.src_r0 = 0; t.src_c0 = 0;
    t.M = m; t.N = pn; t.k0 = c0; t.klen = pn; t.dinv_off = slot; t.dinv_ld = pn;
    slot += (int64_t)pn * pn;
    push_gemm(B.P, t, t.N > 64 ? 128 : tile_for(t.M, t.N));
  }
  return B.run((hipStream_t)stream, nullptr, nullptr, nullptr);
}

int spllt_update_block_hip(void* stream, int m, int n, double* dest, int is_diag, int n1,
                           const double* src1, const double* src2) {
  if (m <= 0 || n <= 0 || n1 < 0 || !dest || !src1 || !src2) return SPLLT_ERROR_PARAMETER;
  if (n1 == 0) return 0;
  OpsBatch B;
  B.bc_off = {abs_off(src2), abs_off(src1)};
  B.bc_w = {n1, n1};
  UpdUnit u{};
  u.mode = MODE_DIRECT; u.lower = is_diag ? 1 : 0;
  u.d_off = abs_off(dest); u.d_ld = n; u.d_row0 = 0; u.d_col0 = 0;
  u.src_bcol0 = 0; u.nseg = 1; u.seg_r0 = 0; u.seg_stride = 0; u.src_r0 = 0; u.M = m;
  u.b_bcol0 = 1; u.b_seg_r0 = 0; u.src_c0 = 0; u.N = n;
  u.k0 = 0; u.klen = n1;
  push_gemm(B.P, u, tile_for(m, n));
  return B.run((hipStream_t)stream, nullptr, nullptr, nullptr);
}

int spllt_update_between_hip(void* stream, double* dest, int blkn, int n1, const double* csrc,
                             int cls, const double* rsrc, int rls, const int* row_list,
                             const int* col_list, int ndiag) {
  if (!dest || !csrc || !rsrc || !row_list || !col_list || blkn <= 0) return SPLLT_ERROR_PARAMETER;
  if (rls <= 0 || cls <= 0 || n1 <= 0) return 0;
  OpsBatch B;
  B.bc_off = {abs_off(rsrc), abs_off(csrc)};
  B.bc_w = {n1, n1};
  UpdUnit u{};
  u.mode = MODE_SCATTER; u.lower = ndiag > 0 ? 1 : 0;
  u.d_off = abs_off(dest); u.d_ld = blkn; u.d_row0 = 0; u.d_col0 = 0;
  u.relrow_off = 0; u.gcol_off = 0;
  u.src_bcol0 = 0; u.nseg = 1; u.seg_r0 = 0; u.seg_stride = 0; u.src_r0 = 0; u.M = rls;
  u.b_bcol0 = 1; u.b_seg_r0 = 0; u.src_c0 = 0; u.N = cls;
  u.k0 = 0; u.klen = n1;
  push_gemm(B.P, u, tile_for(rls, cls));
  return B.run((hipStream_t)stream, nullptr, row_list, col_list);
}

int spllt_expand_buffer_hip(void* stream, double* a, int blkn, const int* row_list, int rls,
                            const int* col_list, int cls, int ndiag, const double* buffer) {
  if (!a || !row_list || !col_list || !buffer) return SPLLT_ERROR_PARAMETER;
  launch_expand_buffer((hipStream_t)stream, a, blkn, row_list, rls, col_list, cls, ndiag, buffer);
  return hipGetLastError() == hipSuccess ? 0 : SPLLT_ERROR_HIP;
}

int spllt_scatter_block_hip(void* stream, int s_m, int s_n, const int* rsrc_index,
                            const int* csrc_index, const double* src, int lds,
                            const int* rdest_index, int d_m, const int* cdest_index, int d_n,
                            double* dest, int ldd) {
  if (!rsrc_index || !csrc_index || !src || !rdest_index || !cdest_index || !dest)
    return SPLLT_ERROR_PARAMETER;
  launch_scatter_block((hipStream_t)stream, s_m, s_n, rsrc_index, csrc_index, src, lds, rdest_index,
                       d_m, cdest_index, d_n, dest, ldd);
  return hipGetLastError() == hipSuccess ? 0 : SPLLT_ERROR_HIP;
}

int spllt_init_lfact_hip(void* stream, double* L, const double* val, const int64_t* dst,
                         const int64_t* src, int64_t n) {
  if (!L || !val || !dst || !src) return SPLLT_ERROR_PARAMETER;
  launch_scatter_val((hipStream_t)stream, L, val, dst, src, n);
  return hipGetLastError() == hipSuccess ? 0 : SPLLT_ERROR_HIP;
}

}  // extern "C"
