// Stream-DAG program of one factorization: the host-side replacement for
// SpLLT's task submission layer (reference src/spllt_factorization_mod.F90:474-751
// and src/spllt_factorization_task_mod.F90).  Instead of one runtime task per
// tile kernel, the elimination tree is cut into levels and every level is a
// short sequence of *batched* launches whose work lists live in device memory:
//
//   per level, per block-column step c, per inner panel p (width <= PW):
//       POTRF  : diagonal panel blocks      (factorize_block, kernels_mod:1168)
//       TRSM   : rows below each panel      (solve_block,     kernels_mod:1217)
//       UPDATE : rest of the block column   (update_block,    kernels_mod:1261)
//     UPDATE   : trailing block columns of the same node with K = blkn
//   UPDATE/scatter: every (node, ancestor block column) pair of the level
//                (update_between + expand_buffer, kernels_mod:2108, :2010)
//
// Everything here is plain C++ (no HIP) so that it can be unit-tested on CPU.
#pragma once
#include <cstdint>
#include <vector>

#include "symbolic.hpp"

namespace spx {

constexpr int kPanelMax = 64;  // widest diagonal panel the POTRF kernel accepts

enum UnitMode : int { MODE_DIRECT = 0, MODE_SCATTER = 1, MODE_TRSM = 2 };

// One batched-GEMM work unit:   C[rowmap(i)][colmap(j)] (-)= sum_seg A_seg[i][:] . B_seg[j][:]
// A rows are node-local rows [src_r0, src_r0+M) of the source supernode, B rows
// are [src_c0, src_c0+N); K runs over `nseg` consecutive block columns of the
// source (or over columns [k0, k0+klen) of a single one).  Only entries with
// src_r0+i >= src_c0+j (the lower triangle in source-row terms) are written.
struct UpdUnit {
  int64_t d_off;       // arena offset of the destination block column
  int64_t relrow_off;  // SCATTER: offset into relpos[] of the entry for i = 0
  int64_t gcol_off;    // SCATTER: offset into rlist[] of the entry for j = 0
  int64_t dinv_off;    // TRSM: offset of the inverted diagonal panel in the dinv scratch
  int src_bcol0;       // first source block column (global id)
  int nseg;            // number of source block columns (K segments)
  int seg_r0;          // node-local row of the first stored row of segment 0
  int seg_stride;      // nb (each following segment starts nb rows lower)
  int src_r0, src_c0;  // node-local first A / B row
  int M, N;
  int k0, klen;        // nseg == 1: column window inside the block column (klen < 0: all)
  int d_ld;            // destination row width
  int d_row0;          // DIRECT/TRSM: stored row of i = 0;  SCATTER: node-local r0 of dest bcol
  int d_col0;          // DIRECT/TRSM: column of j = 0;      SCATTER: pivot position of dest col 0
  int mode;
  int dinv_ld;
  int lower;           // 1: write only entries with src_r0+i >= src_c0+j
  int b_bcol0;         // >= 0: B rows come from this block column (else same as A)
  int b_seg_r0;        // node-local row of the first stored row of B's segment 0
  int atomic;          // DIRECT: 1 = subtract with atomics (another launch or unit may update the
                       // same entries concurrently), 0 = plain read-modify-write (exclusive owner)
  int a_w;             // width and arena offset of source block column src_bcol0 (segment 0):
  int64_t a_off;       // saves the kernel one dependent table lookup before its first loads
};
static_assert(sizeof(UpdUnit) == 120, "UpdUnit layout (mirrored in spllt_amd/api.py)");

struct UpdTile {
  int unit;
  short ti, tj;
};

struct PotrfUnit {
  int64_t off;       // arena offset of the panel's (0,0) entry
  int64_t dinv_off;  // where the inverse of the factored panel goes
  int ld;            // row width of the block column
  int n;             // panel order (<= kPanelMax)
  int gcol;          // pivot position of the panel's first column (error reporting)
  int flags;         // bit 0: block is already a Cholesky factor, only invert it
};

// All sub-diagonal rows of one block column: X = A * inv(L_tile)^T by blocked
// substitution over the block column's panels inside ONE kernel (k_trsm_strip).
struct StripUnit {
  int64_t off;       // arena offset of the block column
  int64_t dinv_off;  // dinv slot of panel 0 (the panels' slots are consecutive)
  int ld;            // block column width w
  int row0;          // first stored row of the region (= w: the rows below the diagonal tile)
  int nrows;         // rows in the region
  int pw;            // panel width
};

// One panel step of the chain below its POTRF, fused (k_panel_step): the rows
// below panel p of a block column are solved against inv(L_pp) and the NEXT
// 64-wide panel of the node (same block column, or panel 0 of the next block
// column) receives every update that is still missing: the previous block
// column (s_*), the source block column's panels before p (global) and panel p
// itself (straight from LDS).  Row coordinates are stored rows of the source
// block column.
struct PanelStepUnit {
  int64_t off;       // arena offset of the source block column
  int64_t dinv_off;  // inv(L_pp) in the dinv scratch (row-major, ld = pn)
  int64_t d_off;     // arena offset of the destination block column (-1: no next panel)
  int64_t s_off;     // arena offset of the previous block column (-1: none)
  int ld;            // source block column width
  int c0, pn;        // panel p: first column, width
  int nrows;         // rows below the panel's diagonal block
  int d_ld, d_c0, d_pn;  // destination row width, first column, panel width
  int d_rshift;      // destination stored row = source stored row - d_rshift
  int s_ld, s_k;     // previous block column: row width, K extent
  int s_rshift;      // previous block column stored row = source stored row + s_rshift
  int pad_;
};

enum LaunchKind : int { L_POTRF = 0, L_GEMM = 1, L_EXCHANGE = 2, L_STRIP = 3, L_CHAIN = 4, L_PANEL = 5 };

struct Launch {
  int kind;
  int level;
  int64_t first, count;  // range in potrf_units (L_POTRF) or tiles (L_GEMM)
  int tile;              // L_GEMM: tile edge (128 or 64 ...)
  double flops;          // useful flops of this launch (for reporting)
  // stream-DAG edges: the launch goes to `stream` (0 = panel stream: POTRF/TRSM
  // chain, 1 = bulk stream: trailing and inter-node updates) after waiting for
  // events wait0/wait1 (-1 = none) and records event `record` (-1 = none).
  int stream = 0;
  int wait0 = -1, wait1 = -1;
  int record = -1;
  int overlap = 0;  // 1: bulk launch that runs beside a panel chain (engine may cap its CU share)
};

struct Program {
  int pw = 64;  // inner panel width
  std::vector<PotrfUnit> potrf_units;
  std::vector<PanelStepUnit> panel_units;  // L_PANEL (tiles: unit, ti = 32-row tile)
  std::vector<PotrfUnit> chain_units;  // L_CHAIN: off = block column, n = tile order, flags = panel width
  std::vector<UpdUnit> units;
  std::vector<UpdTile> tiles;
  std::vector<StripUnit> strip_units;  // L_STRIP launches: tiles[].unit indexes this, .ti = strip
  std::vector<Launch> launches;
  std::vector<int> relpos;      // per (node, touched ancestor): positions of the node's rows in the ancestor's row list
  int64_t dinv_size = 0;        // doubles
  int nevents = 0;              // number of distinct event ids used by the launches
  int final_event = -1;         // recorded (on the bulk stream) when everything is done
  double flops_potrf = 0, flops_trsm = 0, flops_update = 0, flops_between = 0;
};

struct ScheduleOptions {
  int pw = 64;          // inner panel width (<= kPanelMax)
  int tile = 128;       // GEMM tile edge for large units
  // multi-GPU subtree partition: node_owner[s] = owning rank of a pruned-subtree
  // node, -1 for the (replicated) top tree.  With nranks > 1 the program is
  // [own subtrees] EXCHANGE [top tree].
  int rank = 0;
  int nranks = 1;
  const int* node_owner = nullptr;
  bool lookahead = true;  // two-stream schedule: panel chain of block column c+1
                          // overlaps the trailing update by block column c
  bool lazy_next = false;  // (lookahead, unfused) the update c -> c+1 is merged, panel by panel,
                          // into the left-looking update launches of block column c+1
  bool panel_step = false;  // fused TRSM + next-panel update launches (k_panel_step) on levels
  int panel_step_limit = 768;  // ... whose steps have at most this many 32-row tiles
  bool slice_between = true;  // (lookahead) inter-node updates are issued in K slices on a third
  int slice_width = 2;        // stream while the panel chains of the level are still running
  bool fused_strip = false;  // sub-diagonal rows of a block column in one k_trsm_strip launch
  int strip_limit = 512;    // ... on levels whose steps have at most this many strips
  bool tile_chain = true;   // with fused_strip: the panel chain of a diagonal tile (w <= 256)
                            // runs as ONE workgroup (k_tile_chain) instead of 3*np-1 launches
};

void build_program(const Symbolic& S, const ScheduleOptions& opt, Program& P);

// ---------------------------------------------------------------------------
// Triangular solves with the device-resident factor (SURVEY.md 8(f) row f2;
// reference solve_fwd / solve_bwd, src/spllt_solve_mod.F90:244-411).  The same
// level batching as the factorization: per level and block-column step one
// "diag" launch (solve with the diagonal tile through the inverted 64x64
// panels the factorization left in the dinv scratch) and one "strip" launch
// (rows below: forward  y[rows] -= L_strip x,  backward  y[cols] -= L_strip^T x[rows]).
// ---------------------------------------------------------------------------
struct SolveUnit {
  int64_t off;       // arena offset of the block column
  int64_t dinv_off;  // dinv slot of its panel 0
  int64_t idx_off;   // offset into rlist[] of the block column's first row (= its first column)
  int w;             // width
  int nrow;          // rows stored (w diagonal rows + rows below)
  int pw;            // panel width
  int pad_;
};

enum SolveKind : int { SV_DIAG_FWD = 0, SV_STRIP_FWD = 1, SV_STRIP_BWD = 2, SV_DIAG_BWD = 3 };

struct SolveLaunch {
  int kind;
  int level;
  int64_t first, count;  // DIAG: range in units; STRIP: range in tiles (unit, strip)
};

constexpr int kSolveStripRows = 64;

struct SolveProgram {
  std::vector<SolveUnit> units;   // one per block column, indexed by block column id
  std::vector<int> diag_list;     // unit ids in launch order (DIAG launches index this)
  std::vector<UpdTile> tiles;     // (unit, strip) pairs in launch order
  // fwd = [own subtrees..., top tree...], bwd = its mirror [top tree..., own subtrees...].
  // Without a partition everything counts as "own subtrees" (fwd_nsub = fwd.size(),
  // bwd_ntop = 0).  A partitioned solve runs in three phases with an all-reduce of
  // the right-hand side vector after the first and after the last:
  //   0: fwd[0, fwd_nsub)   1: fwd[fwd_nsub, end) + bwd[0, bwd_ntop)   2: bwd[bwd_ntop, end)
  std::vector<SolveLaunch> fwd, bwd;
  size_t fwd_nsub = 0, bwd_ntop = 0;
};

// node_owner (per node: owning rank, -1 = top tree) may be null: no partition
void build_solve_program(const Symbolic& S, int pw, SolveProgram& P, const int* node_owner = nullptr,
                         int rank = 0);

}  // namespace spx
