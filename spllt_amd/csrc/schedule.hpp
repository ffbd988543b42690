// Stream-DAG program of one factorization: the host-side replacement for
// SpLLT's task submission layer (reference src/spllt_factorization_mod.F90:474-751
// and src/spllt_factorization_task_mod.F90).  Instead of one runtime task per
// tile kernel, the elimination tree is cut into levels and every level is a
// short sequence of *batched* launches whose work lists live in device memory:
//
//   per level, per block-column step c, per panel p (<= PW columns):
//       CHAIN  : k_chain_potrf, one workgroup per node: POTRF of the panel's diagonal block
//                (factorize_block, kernels_mod:1168) + its inverse
//       TRSM   : rows below the panel (solve_block, kernels_mod:1217) as a product with the inverse
//       UPDATE : left-looking update of the next panel's columns (update_block, :1261)
//       (or PANEL: all three in one k_panel launch, when the step has few row blocks)
//     UPDATE   : trailing block columns of the same node
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

enum UnitMode : int { MODE_DIRECT = 0, MODE_SCATTER = 1, MODE_TRSM = 2, MODE_BUFFER = 3, MODE_GEN = 4 };
// MODE_GEN (subtree tasks, L_SUBTREE): the part of a node's update that LEAVES its subtree is added
// into the subtree's generated element -- the lower triangle, packed by rows, of the square matrix
// over the rows below the subtree root's columns (d_off = its offset in the generated-element
// scratch; row i of it starts at i (i + 1) / 2) -- at the positions relpos[relrow_off + i] (row)
// and relpos[gcol_off + j] (column) of the root's row list.
// MODE_BUFFER (deterministic engine): the product of an inter-node update unit is STORED, as a
// dense M x N row-major block, in a scratch buffer (d_off = its offset there, d_ld = N); a
// k_gather launch then subtracts the buffered blocks from their destination tiles in a fixed
// order -- the reference's own two steps, update_between into a buffer + expand_buffer
// (kernels_mod:2108-2237, :2010-2053), made destination-centric instead of atomic.

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

// One step of the panel chain (k_chain_potrf), one workgroup: panel [c0, c0+pn) of a block
// column.  winv_off: where inv(L_pp) goes in the dinv scratch (pn x pn; cs = c0, ce = c0 + pn:
// the fields of the removed wider "chain block" sub-tiles).
struct ChainUnit {
  int64_t off;       // arena offset of the block column
  int64_t winv_off;
  int ld;            // block column width
  int c0, pn;        // panel: first column, width (<= kPanelMax)
  int cs, ce;        // sub-tile: columns (= stored rows) [cs, ce)
  int gcol;          // pivot position of column c0 (error reporting)
};
static_assert(sizeof(ChainUnit) == 40, "ChainUnit layout (mirrored in spllt_amd/api.py)");

// One whole panel step of a block column (k_panel): POTRF of the panel [c0, c0+pn), solve of
// all rows below it, left-looking update of the next panel's columns (next_pn of them; 0: the
// panel is the last of its block column).  The inverse goes to dinv_off (pn x pn).
struct PanelUnit {
  int64_t off;       // arena offset of the block column
  int64_t dinv_off;
  int ld;            // block column width
  int c0, pn;
  int next_pn;
  int nrow;          // rows of the block column
  int gcol;          // pivot position of column c0 (error reporting)
  int ntile;         // workgroups of the unit (64-row blocks below the panel, at least one)
  int pad_;          // flags: bit 0 = the panel is already factored and inverted (a chain launch ran
                     // before): the workgroups load inv(L_pp) from the dinv scratch instead
};
static_assert(sizeof(PanelUnit) == 48, "PanelUnit layout (mirrored in spllt_amd/api.py)");

// A small subtree as one device task (L_SUBTREE): nodes [node_first, node_first + node_count) of
// sub_nodes, children before parents, the last one is the subtree's root.
struct SubTask {
  int64_t g_off;     // offset of the generated element in the scratch (packed lower triangle)
  int node_first, node_count;
  int g_n;           // its order = rows below the root's columns (0: a single node, nothing to collect)
  int pad_;
};
static_assert(sizeof(SubTask) == 24, "SubTask layout (mirrored in spllt_amd/api.py)");
struct SubNode {
  int64_t off;       // arena offset of the node's block column (it has one, <= one panel wide)
  int64_t dinv_off;  // where the inverse of its diagonal block goes
  int w, nrow;       // columns, rows
  int gcol;          // pivot position of column 0 (error reporting)
  int unit_first, unit_count;   // its update units (units[]: MODE_SCATTER / MODE_GEN, K = all w columns)
  int root;          // 1: the subtree's root -- its units add the generated element to what they scatter
};
static_assert(sizeof(SubNode) == 40, "SubNode layout (mirrored in spllt_amd/api.py)");

// Deterministic assembly (k_gather): one workgroup per destination tile (<= 64 x 64 entries of
// a block column) walks its items in order; an item is the part of one buffered update block
// (MODE_BUFFER unit) that lands in the tile.
struct GatherItem {
  int64_t buf_off;     // scratch offset of the unit's block (entry (0, 0))
  int64_t relrow_off;  // relpos[] offset of the unit's row 0
  int64_t gcol_off;    // rlist[] offset of the unit's column 0
  int ld;              // row width of the buffered block (= unit N)
  int i0, i1, j0, j1;  // rows / columns of the block that land in the tile
  int diag_shift;      // src_r0 - src_c0 of the unit: only entries with diag_shift + i >= j count
  int lower, pad_;
};
static_assert(sizeof(GatherItem) == 56, "GatherItem layout (mirrored in spllt_amd/api.py)");
struct GatherTile {
  int64_t d_off;       // arena offset of the destination block column
  int d_ld;            // its row width
  int row0, col0;      // tile origin (stored row / column of the block column)
  int rows, cols;      // tile extent (<= 64 each)
  int drow_base;       // node-local row of the block column's stored row 0 (relpos is node-local)
  int dcol_base;       // pivot position of the block column's column 0 (rlist holds pivot positions)
  int first, count;    // range in gather_items
  int pad_;
};
static_assert(sizeof(GatherTile) == 48, "GatherTile layout (mirrored in spllt_amd/api.py)");

enum LaunchKind : int { L_POTRF = 0, L_GEMM = 1, L_EXCHANGE = 2, L_CHAIN = 4, /* 5: removed */ L_GATHER = 6,
                        L_PANEL = 7, L_CHAIN4 = 8, L_TRSM4 = 9, L_SUBTREE = 10 };
// L_SUBTREE (k_subtree): one workgroup per SubTask factorizes a whole small subtree, node by node in
// post-order -- the reference's subtree task (a20-a25: spllt_subtree_factorize,
// src/spllt_factorization_mod.F90:196-261, kernels_mod:780-821): Cholesky of the node's (one-panel)
// block column, solve of its rows, then its update units: into the nodes of the subtree directly
// (MODE_SCATTER units), into the subtree's generated element what leaves it (MODE_GEN); the root's
// units carry the generated element with them into the ancestors (ONE extend-add per subtree,
// factorization_mod:39-191).
// L_CHAIN4 (k_chain_block): one workgroup factors the whole diagonal block of a CHAIN BLOCK of up to
// four panels (ChainUnit with pn = its width <= 4 pw) and emits the panels' inverses (per panel, the
// layout of the one-panel chain steps).
// L_TRSM4 (k_trsm_rows): the rows below a chain block solved against it, one workgroup per 64 rows and
// ALL the block's columns (tiles: unit, ti; TRSM-mode UpdUnits with N = K = the block's width).  The
// two replace, per chain block, up to four POTRF, four TRSM and three in-panel update launches of the
// chain stream.  (Round 3's two-panel chain blocks -- k_chain_potrf2, which emitted a 128 x 128
// inverse, and k_trsm2 -- measured equal to the one-panel steps and are gone.)

// streams of the program: the chain (panel chain kernels and the updates that gate them),
// (the side stream id is reserved: a variant that ran the rows below the sub-tiles one step
// behind the chain on their own queue was measured slower and removed), the bulk and
// far streams (trailing / early inter-node updates that run BESIDE a chain: the engine masks
// them off a few CUs so that chain and side kernels always find a free CU) and the wide stream
// (launches that have the chip to themselves: no mask)
enum StreamId : int { ST_CHAIN = 0, ST_BULK = 1, ST_FAR = 2, ST_SIDE = 3, ST_WIDE = 4, ST_COUNT = 5 };

struct Launch {
  int kind;
  int level;
  int64_t first, count;  // range in chain_units (L_CHAIN), potrf_units (L_POTRF) or tiles (L_GEMM)
  int tile;              // L_GEMM: tile edge (128, 64 or 32); L_CHAIN: most rows below a panel
  double flops;          // useful flops of this launch (for reporting)
  // stream-DAG edges: the launch goes to `stream` (StreamId) after waiting for the events
  // wait[0..3] (-1 = none) and records event `record` (-1 = none).
  int stream = 0;
  int wait[4] = {-1, -1, -1, -1};
  int record = -1;
  int overlap = 0;  // 1: bulk launch that runs beside a panel chain (engine may cap its CU share)
  int lat = 0;      // 1: a step of the panel chain (TRSM, in-panel update, update of the next block
                    // column): few tiles, the next step waits for it -- the engine may pick a
                    // kernel variant built for latency instead of throughput
  void add_wait(int ev) {
    if (ev < 0) return;
    for (int& w : wait) {
      if (w == ev) return;
      if (w < 0) { w = ev; return; }
    }
  }
};

// Points of a partitioned (multi-GPU) program where the ranks exchange block columns of the top
// tree through the exchange buffer (host: a collective on the engine's stream between the pack and
// the unpack).  Every rank's program has the same exchanges in the same order.
enum ExchangeKind : int {
  X_REDUCE_ALL = 0,    // all-reduce(sum) of every top-tree block column + one indicator element
                       // (replicated top tree: everybody continues with the whole sum)
  X_REDUCE_OWNER = 1,  // reduce-scatter(sum): rank r's chunk [r*chunk, (r+1)*chunk) holds the block
                       // columns r owns (distributed top tree)
  X_BCAST = 2,         // the owners broadcast the block columns of a finished step (one segment of
                       // the buffer per root, items of a root contiguous)
  X_FLAG = 3           // all-reduce(sum) of the one-element "not positive definite" indicator
};
struct ExchangeItem {
  int bcol;
  int root;        // owner (X_REDUCE_ALL: -1)
  int64_t xoff;    // offset in the exchange buffer
  int64_t count;   // doubles: nrow * width (space 0), the block column's dinv slots (space 1)
  int64_t off;     // offset in the arena (space 0) / the dinv scratch (space 1)
  int space;       // 0: arena; 1: dinv scratch -- a broadcast block column travels with the
                   // inverses of its diagonal panels, which the solve needs on every rank
};
struct Exchange {
  int kind;
  int first_item, nitems;   // range in xitems
  int64_t elems;            // doubles of the buffer the collective covers (from offset 0)
  int64_t chunk;            // X_REDUCE_OWNER: elems / nranks
};

struct Program {
  int pw = 64;  // inner panel width
  int cb = 64;  // = pw (the chain block of the removed sub-tile chain kernels; layout parameter of the dinv slots)
  std::vector<PotrfUnit> potrf_units;  // L_POTRF (operator twins only: inverse of given factors)
  std::vector<ChainUnit> chain_units;  // L_CHAIN
  std::vector<PanelUnit> panel_units;  // L_PANEL (tiles: unit, ti)
  std::vector<SubTask> sub_tasks;      // L_SUBTREE
  std::vector<SubNode> sub_nodes;
  int64_t gen_size = 0;                // doubles of the generated-element scratch (zero between factorizations:
                                       // the root's units clear what they read)
  std::vector<GatherItem> gather_items;
  std::vector<GatherTile> gather_tiles;  // L_GATHER
  int64_t scratch_size = 0;     // doubles of the MODE_BUFFER scratch (largest launch)
  std::vector<UpdUnit> units;
  std::vector<UpdTile> tiles;
  std::vector<Launch> launches;
  std::vector<int> relpos;      // per (node, touched ancestor): positions of the node's rows in the ancestor's row list
  int64_t dinv_size = 0;        // doubles
  // L_EXCHANGE launches (multi-GPU): Launch::first = index into exchanges
  std::vector<Exchange> exchanges;
  std::vector<ExchangeItem> xitems;
  int64_t xbuf_elems = 0;       // size the exchange buffer must have (doubles)
  int nevents = 0;              // number of distinct event ids used by the launches
  int final_event = -1;         // recorded when everything is done
  double flops_potrf = 0, flops_trsm = 0, flops_update = 0, flops_between = 0;
};

struct ScheduleOptions {
  int pw = 64;          // inner panel width (<= kPanelMax)
  int tile = 128;       // GEMM tile edge for large units
  int cb = 64;          // ignored (chain block of the removed sub-tile chain kernels)
  bool chain4 = false;  // block-column steps whose block columns are wider than one panel: chain blocks of
                        // up to FOUR panels (L_CHAIN4 + L_TRSM4, two dependent launches per 4 pw columns of the
                        // panel chain instead of twelve); false (SPLLT_CHAIN4=0): one panel per chain step
  // multi-GPU subtree partition: node_owner[s] = owning rank of a pruned-subtree
  // node, -1 for the (replicated) top tree.  With nranks > 1 the program is
  // [own subtrees] EXCHANGE [top tree].
  int rank = 0;
  int nranks = 1;
  const int* node_owner = nullptr;
  // distributed top tree: top_owner[b] = rank that owns block column b of the
  // top tree (-1 elsewhere).  The extend-add at the exchange point goes to the owners only
  // (X_REDUCE_OWNER); in the top tree a block column is factorized by its owner, broadcast
  // (X_BCAST), and every rank applies it to the destination block columns it owns.  null: the top
  // tree is replicated (X_REDUCE_ALL, every rank factorizes all of it).
  const int* top_owner = nullptr;
  bool lookahead = true;  // multi-stream schedule: the chain of block column c+1 overlaps the
                          // trailing update by block column c; false: one stream, program order
  bool slice_between = true;  // (lookahead) inter-node updates are issued in K slices on the far
  int slice_width = 2;        // stream while the panel chains of the level are still running
  bool zones = true;          // (lookahead) the inter-node updates at the end of a level are issued
                              // sorted by destination block column, one event per zone, so that
                              // the next level's chains start beside them; the trailing updates
                              // of a level that starts this way subtract with atomics
  bool fused_panel = true;    // steps with few row blocks below the panels: one k_panel launch per panel
  int fused_panel_max = 64;   // ... when the launch has at most this many workgroups.  (k_panel needs a
                              // whole CU per workgroup; beside the trailing updates only the reserved
                              // CUs are free, and a launch that needs a second round of them loses
                              // what the two saved kernel boundaries gain: 26.3 ms unfused, 25.9 with
                              // 64, 26.3 with 128, 28.4 with 512 on the nd24k stand-in)
  int super_panel = 256;      // block columns wider than this: left-looking panel updates only inside a
                              // super-panel of this many columns, one right-looking update of the rest of
                              // the block column per finished super-panel (0: left-looking throughout)
  int tile128_min = 4096;     // launches of up to this many 64-tiles keep 64-tiles (the 128-tile pays when a
                              // launch fills the chip for several rounds; throughput-bound problems: 1024)
  bool pair_sources = true;   // trailing updates inside a node by two source block columns at a time
  bool subtrees = false;      // small subtrees (every node one panel wide at most, modelled time within
  int subtree_us = 300;       // subtree_us) run as ONE workgroup each (L_SUBTREE) instead of level by level;
                              // single-GPU, non-deterministic programs only.  Off (SPLLT_SUBTREES=1, engine flag
                              // bit 18): measured slower at every budget and size (profiles/r04/subtree_tasks_ab.txt)
                              // -- a task walks its nodes one after the other on one CU, the level-batched
                              // launches spread the nodes of a level, and the tiles of a node, over the chip
  bool deterministic = false; // no atomics: inter-node updates through a buffer + ordered gather
                              // (MODE_BUFFER / k_gather); implies no zones, no early slices
  int buffer_levels = 0;      // the inter-node updates of the tree levels below this one go through the
                              // buffer + ordered gather as well (the lowest levels have narrow sources:
                              // their scatter-adds are bound by the chip's ~1.3 TB/s of fp64 atomics, which
                              // plain stores + one read-modify-write per destination entry are not)
};

// dinv scratch of a block column of width w: per CHAIN BLOCK (cb columns from column g0 = 0, cb,
// 2 cb, ...; the last one ragged) one square cw x cw row-major matrix, cw = min(cb, w - g0): the
// inverse of that diagonal block of L (lower triangle; what lies above the diagonal is zero and is
// never written).  The rows of panel p (first column c0 = p pw) start at winv_offset; the row
// stride is winv_ld.  With cb = pw a chain block is one panel and the slot is the pn x pn inverse.
inline int winv_ld(int w, int cb, int c0) {
  const int g0 = (c0 / cb) * cb;
  return cb < w - g0 ? cb : w - g0;
}
inline int64_t winv_total(int w, int cb) {
  int64_t o = 0;
  for (int g0 = 0; g0 < w; g0 += cb) {
    const int64_t cw = cb < w - g0 ? cb : w - g0;
    o += cw * cw;
  }
  return o;
}
inline int64_t winv_offset(int w, int pw, int cb, int p) {
  const int c0 = p * pw;
  if (c0 >= w) return winv_total(w, cb);
  const int g0 = (c0 / cb) * cb;
  int64_t o = 0;
  for (int t = 0; t < g0; t += cb) {
    const int64_t cw = cb < w - t ? cb : w - t;
    o += cw * cw;
  }
  return o + (int64_t)(c0 - g0) * winv_ld(w, cb, c0);
}

void build_program(const Symbolic& S, const ScheduleOptions& opt, Program& P);

// owners of the top-tree block columns of a partition (node_owner from assign_owners): dealt
// round-robin in the order the top tree is walked (level, step, node), so that the block columns
// of one step - whose panel chains can run at the same time - sit on different ranks, and the
// block columns of a node are spread 1-D cyclically.  -1 for block columns outside the top tree.
void assign_top_owners(const Symbolic& S, const std::vector<int>& node_owner, int nranks,
                       std::vector<int>& top_owner);
// Is a distributed top tree worth its broadcasts?  (flops the replicated top tree repeats on
// every rank against the latency the per-step broadcasts add)
bool distribute_top_tree(const Symbolic& S, const std::vector<int>& node_owner, int nranks);

// Is the factorization of S bound by the latency of its panel chains rather than by matrix
// throughput?  Estimate: the longest chain of dependent panel steps through the tree (one step
// per `pw` columns of the widest node of every level, ~60 us each) against the time the flops
// take at a typical update rate.  Latency-bound problems get CUs reserved for the chain and the
// zone pipeline; throughput-bound ones keep the whole chip for the updates (measured: the
// reservation costs 2-3 % on the 12-48 TFLOP configurations and gains 8 % on the 0.76 TFLOP one).
bool latency_bound(const Symbolic& S, int pw);

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
  int cb;            // chain block (Winv layout, see ChainUnit)
  int gcol0;         // pivot position of column 0: rlist[idx_off + j] = gcol0 + j for j < w (the block column's
  int pad_;          // own columns are consecutive: the kernels skip that index load)
};
static_assert(sizeof(SolveUnit) == 48, "SolveUnit layout (mirrored in spllt_amd/api.py)");

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
void build_solve_program(const Symbolic& S, int pw, int cb, SolveProgram& P,
                         const int* node_owner = nullptr, int rank = 0);

}  // namespace spx
