// Launch wrappers of the gfx950 kernels (kernels.hip).  Every wrapper only
// enqueues work on `st`; none synchronises.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "schedule.hpp"

namespace spx {

// Where a kernel goes: onto a stream (eager), or into a HIP graph as a kernel node behind
// `deps` (graph construction, Engine::build_graph; `node` returns the node, `err` the status).
struct LaunchSink {
  hipStream_t stream = nullptr;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  mutable hipGraphNode_t node = nullptr;
  mutable hipError_t err = hipSuccess;
  LaunchSink() = default;
  LaunchSink(hipStream_t st) : stream(st) {}   // every wrapper below also takes a plain stream
};

void launch_scatter_val(const LaunchSink& st, double* L, const double* val, const int64_t* dst,
                        const int64_t* src, int64_t n);
// the arena cleared and A copied in, in one pass (cptr / loc / src: the val -> L map bucketed by
// chunks of kInitChunk doubles of the arena: entries of chunk c are [cptr[c], cptr[c + 1]), loc = the
// entry's position inside the chunk, src = its index in val)
constexpr int kInitChunk = 4096;
void launch_init_arena(const LaunchSink& st, double* L, int64_t arena, const double* val, const int64_t* cptr,
                       const unsigned short* loc, const int* src);
// (unit0 = host copy of units[0]: travels with the kernel arguments)
void launch_potrf(hipStream_t st, const PotrfUnit* units, int64_t count, double* L, double* dinv,
                  int* flag, const PotrfUnit& unit0);
// one small subtree per workgroup (SubTask): its nodes in post-order, what leaves the subtree through the
// generated-element scratch `gen` (zero before and after)
void launch_subtree(const LaunchSink& st, const SubTask* tasks, int64_t count, const SubNode* nodes,
                    const UpdUnit* units, const int* relpos, const int* rlist, double* L, double* dinv, double* gen,
                    int* flag);
// one step of the panel chain per workgroup (ChainUnit): POTRF of the panel + its inverse
void launch_chain_panel(const LaunchSink& st, const ChainUnit* units, int64_t count, double* L, double* dinv,
                        int* flag, const ChainUnit& unit0);
// one chain block of up to four panels per workgroup (ChainUnit, pn = its width <= 4 pw): the whole
// diagonal block factored, the panels' inverses emitted
void launch_chain_block(const LaunchSink& st, const ChainUnit* units, int64_t count, double* L, double* dinv,
                        int* flag, int pw, const ChainUnit& unit0);
// the rows below a chain block solved against it, 64 rows x all columns per workgroup (tiles: unit, ti)
void launch_trsm_rows(const LaunchSink& st, const UpdTile* tiles, int64_t count, const UpdUnit* units, double* L,
                      const double* dinv, int pw, int prio);
// one whole panel step per launch (PanelUnit; tiles: unit, ti = 64-row block below the panel)
// counters: two zero-initialised ints per panel unit (left zero again by the launch)
void launch_panel(const LaunchSink& st, const UpdTile* tiles, int64_t count, const PanelUnit* units, double* L,
                  double* dinv, int* counters, int* flag);
// deterministic assembly of buffered inter-node update blocks (GatherTile / GatherItem)
void launch_gather(const LaunchSink& st, const GatherTile* tiles, int64_t count, const GatherItem* items,
                   double* L, const double* scratch, const int* relpos, const int* rlist);
// multi-GPU: not-positive-definite flag <-> extra element of the exchange buffer
void launch_flag_pack(hipStream_t st, const int* flag, double* slot);
void launch_flag_unpack(hipStream_t st, const double* slot, int* flag);
// y[q * ldy + i] *= keep[i]  (multi-GPU solve: a rank's share of a distributed vector)
void launch_mask(hipStream_t st, double* y, const double* keep, int n, int nrhs, int64_t ldy);
// debug: fill the LDS of every CU with signalling NaNs
void launch_poison_lds(hipStream_t st);
void launch_update(const LaunchSink& st, int tile, const UpdTile* tiles, int64_t count,
                   const UpdUnit* units, const int64_t* bc_off, const int* bc_w, double* L,
                   const int* relpos, const int* rlist, const double* dinv, int prio = 0,
                   int lds_pad = 0, bool allow_dma = true, bool latency = false);
// (latency: a small launch on the critical path -- 64 columns of K per LDS step instead of 16, so
// that a tile makes a quarter of the dependent round trips to memory)
// (allow_dma = false: the register-staged kernels -- for operands in caller-owned buffers without
// slack behind them: the DMA kernels load whole 16-column chunks)
void launch_scatter_block(hipStream_t st, int s_m, int s_n, const int* rsrc_index,
                          const int* csrc_index, const double* src, int lds,
                          const int* rdest_index, int d_m, const int* cdest_index, int d_n,
                          double* dest, int ldd);

// one launch of the device solve (kind = SolveKind); four: every block column of a DIAG launch has at
// most four 64-wide panels (pw = cb = 64, w <= 256) -- the kernel that reads L in one round trip
void launch_solve(hipStream_t st, int kind, const int* list, const UpdTile* tiles, int64_t first,
                  int64_t count, const SolveUnit* units, const double* L, const double* dinv,
                  const int* rlist, double* y, int nr, int64_t ldy, bool four = false,
                  const SolveUnit* one = nullptr);   // one: the launch works on ONE block column (host copy of its unit;
                                                     // strips: strip i = workgroup i) -- the descriptor travels with the arguments
void launch_expand_buffer(hipStream_t st, double* a, int blkn, const int* row_list, int rls,
                          const int* col_list, int cls, int ndiag, const double* buffer);

}  // namespace spx
