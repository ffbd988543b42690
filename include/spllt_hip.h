/*
 * spllt_hip.h -- additions of libspllt_hip.so next to the drop-in ABI of
 * spllt_iface.h.  Two groups:
 *
 *  (1) the per-kernel operator API.  The reference already exposes its factor
 *      kernels as bind(C) routines for the StarPU/PaRSEC task bodies
 *      (src/spllt_kernels_mod.F90:1193,1233,1295,2055,2241 and the CUDA launcher
 *      src/StarPU/expand_buffer_kernels.cu:48-62).  These are their
 *      stream-taking twins: same argument meaning, device pointers, plain
 *      index arrays instead of Fortran derived-type handles, asynchronous on
 *      the given HIP stream (passed as void* so that the header needs no HIP).
 *
 *  (2) engine control / introspection used by benchmarks and tests
 *      (exact 64-bit statistics, device-resident factorization, L download,
 *      export of the symbolic structure and of the stream-DAG program).
 */
#ifndef SPLLT_HIP_H
#define SPLLT_HIP_H
#include <stdint.h>

#include "spllt_iface.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) kernel operators: all pointers are DEVICE pointers ---------------- */

/* twin of spllt_factor_diag_block_c(m, n, bc) (kernels_mod:1193): Cholesky of
 * the n x n head of a row-major m x n diagonal tile and the triangular solve
 * of its m-n trailing rows.  *dev_flag (int, device) receives min(column+1) of
 * a non-positive pivot, untouched otherwise (initialise it to INT_MAX). */
int spllt_factor_diag_block_hip(void *stream, int m, int n, double *bc, int *dev_flag);

/* twin of spllt_solve_block_c(m, n, bc_kk, bc_ik) (kernels_mod:1233):
 * bc_ik <- bc_ik * L_kk^-T, bc_ik is m x n row-major, bc_kk the factored n x n tile. */
int spllt_solve_block_hip(void *stream, int m, int n, const double *bc_kk, double *bc_ik);

/* twin of spllt_update_block_c(m, n, dest, isDiag, n1, src1, src2)
 * (kernels_mod:1295): dest -= src2 * src1^T; src1 is n x n1, src2 is m x n1. */
int spllt_update_block_hip(void *stream, int m, int n, double *dest, int is_diag, int n1,
                           const double *src1, const double *src2);

/* twin of spllt_update_between_c (kernels_mod:2241) with the index lists of
 * spllt_update_between_compute_map_c (:1725) passed explicitly and the
 * spllt_expand_buffer_c step (:2055) fused into the GEMM epilogue:
 *   dest[row_list[i]*blkn + col_list[j]] -= sum_k rsrc[i][k] * csrc[j][k],
 * i < rls, j < (i < ndiag ? i+1 : cls); csrc is cls x n1, rsrc is rls x n1,
 * row_list/col_list are 0-based device arrays. */
int spllt_update_between_hip(void *stream, double *dest, int blkn, int n1, const double *csrc,
                             int cls, const double *rsrc, int rls, const int *row_list,
                             const int *col_list, int ndiag);

/* twin of spllt_expand_buffer_c (kernels_mod:2055) / spllt_cu_expand_buffer
 * (StarPU/expand_buffer_kernels.cu:48): a[row_list[j]*blkn + col_list[i]] +=
 * buffer[j*cls + i], i < (j < ndiag ? j+1 : cls). */
int spllt_expand_buffer_hip(void *stream, double *a, int blkn, const int *row_list, int rls,
                            const int *col_list, int cls, int ndiag, const double *buffer);

/* twin of spllt_scatter_block (kernels_mod:1122): dest -= src with row/column
 * positions located in the destination index lists (all sorted, device). */
int spllt_scatter_block_hip(void *stream, int s_m, int s_n, const int *rsrc_index,
                            const int *csrc_index, const double *src, int lds,
                            const int *rdest_index, int d_m, const int *cdest_index, int d_n,
                            double *dest, int ldd);

/* twin of spllt_init_node_c / spllt_init_blk_c (kernels_mod:2367, :2425) for
 * the whole arena: L[dst[i]] = val[src[i]], i < n (L zeroed by the caller). */
int spllt_init_lfact_hip(void *stream, double *L, const double *val, const int64_t *dst,
                         const int64_t *src, int64_t n);

/* ---- (1b) matrix file readers ------------------------
 * What the reference keeps beside the path for its drivers: MatrixMarket coordinate files
 * (src/spllt_mod.F90:426-491 mm_double_read + :543-620 coo_to_csc_double) and Rutherford-Boeing
 * files of assembled symmetric matrices (SPRAL rb_read as the drivers call it,
 * drivers/spllt_omp.F90:78-85).  Both return the LOWER triangle as 1-based CSC with sorted
 * rows -- the arguments of spllt_analyse / spllt_factor -- in malloc'ed arrays that
 * spllt_hip_free_matrix releases.  values: 0 = as in the file (a pattern-only file is an error),
 * 3 = the drivers' rb_options%values = 3: off-diagonal values as in the file, or made up
 * (uniform in (-1, 1), splitmix64 from `seed`; the reference: SPRAL random_real,
 * spllt_mod.F90:480-485) for a pattern-only file, every diagonal entry 1 + the sum of the
 * |off-diagonal entries| of its row.  A general MatrixMarket matrix is read as (A + A^T) / 2.
 * Returns 0 or an SPLLT error flag (message on stderr). */
int spllt_hip_read_rb(const char *path, int values, int seed, int *n, int *nnz, int **ptr, int **row,
                      double **val);
int spllt_hip_read_mm(const char *path, int values, int seed, int *n, int *nnz, int **ptr, int **row,
                      double **val);
void spllt_hip_free_matrix(int *ptr, int *row, double *val);

/* ---- (2) engine control / introspection ----------------------------------- */

typedef struct {
  int64_t n, nnz_a, nnodes, nbcol, nblk, arena, nnz_l, flops, rlist_len;
  int nb, maxmn, maxdepth, nlevels;
  char ordering[16];
} spllt_hip_sym_info_t;

/* analyse with a caller-supplied pivot order (order_in[i] = 1-based position
 * of variable i; NULL = built-in nested dissection).  Otherwise identical to
 * spllt_analyse. */
void spllt_hip_analyse_ordered(void **akeep, void **fkeep, spllt_options_t *options, int n,
                               const int *ptr, const int *row, spllt_inform_t *info, int *order,
                               const int *order_in);

/* analyse from a complete symbolic factorization: the quintuple SpLLT's own spllt_analyse
 * takes from SSIDS (akeep%sptr, %sparent, %rptr, %rlist and the pivot order, reference
 * src/spllt_analyse_mod.F90:129-158), all 1-based exactly as SSIDS delivers them: sptr
 * (nnodes+1), sparent (nnodes, virtual root = nnodes+1), rptr (nnodes+1, 64-bit), rlist
 * (pivot positions, sorted per node, the node's own columns first), order_in[i] = position
 * of variable i.  The factorization then uses exactly SSIDS' supernode partition and
 * assembly tree (no ordering, supernode detection or amalgamation of our own), so a
 * reference build that keeps SPRAL/Metis for the analyse gets the same tree on both paths.
 * Invalid input (not a permutation, nodes not postordered, unsorted row lists, pattern of A
 * not covered) -> info->flag = SPLLT_ERROR_PARAMETER. */
void spllt_hip_analyse_symbolic(void **akeep, void **fkeep, spllt_options_t *options, int n,
                                const int *ptr, const int *row, spllt_inform_t *info, int nnodes,
                                const int *sptr, const int *sparent, const int64_t *rptr,
                                const int *rlist, const int *order_in);

int spllt_hip_sym_info(const void *akeep, spllt_hip_sym_info_t *out);
/* copy a named 0-based array of the symbolic structure; returns its length in
 * elements (call with buf = NULL to query).  int32 arrays: "order", "sptr",
 * "sparent", "rlist", "small", "level", "bcol_node", "bcol_width", "bcol_r0",
 * "bcol_nrow"; int64 arrays: "rptr", "bcol_off", "map_dst", "map_src",
 * "lmap_ptr", "weight". */
int64_t spllt_hip_sym_get(const void *akeep, const char *name, void *buf, int64_t capacity);

/* engine knobs (before the first spllt_factor on this fkeep); flags: bit 0 =
 * reserved, bit 1 = single-stream program (no lookahead, program order),
 * bit 6 = issue the inter-node updates only at the end of each level (default:
 * in K slices on a separate stream while the level's panel chains still run),
 * bit 7 = debug: the LDS of every CU is filled with signalling NaNs before every
 * kernel launch (a kernel that reads LDS it has not written then computes NaNs),
 * bit 8 = no CU reservation (the default since round 4; SPLLT_HIP_RESERVE_CUS=32 masks the
 * streams that carry the trailing updates off the last 32 CUs, which the latency-critical
 * panel-chain kernels then find free: 0.6 % on the bench workload, and CU-masked streams are
 * what rocprofv3 crashes on at exit).  Bit 9 = no fused panel launches: by default a panel step whose block columns have
 * few rows below the panel (at most 64 blocks of 64 rows in the launch) runs as ONE kernel -
 * every workgroup factors the 64 x 64 diagonal block itself, solves its own rows and applies
 * the left-looking update of the next panel's columns to them - instead of a POTRF, a TRSM
 * and an update launch (used when the chain block is one panel, the default).
 * Bit 10 / bit 11 = force the zone pipeline on / off (inter-node updates at the end
 * of a level issued by destination block column so that the next level's panel chains
 * start beside them; default: on for latency-bound problems, see schedule.hpp).
 * Bit 12 = deterministic engine: no atomic adds anywhere in the factorization.  The
 * inter-node updates store their products in a scratch buffer and one workgroup per
 * destination tile subtracts them in a fixed order (the reference's buffer + expand_buffer
 * steps, serialised per destination like its OpenMP path, task_mod:1239-1241); two
 * factorizations of the same values then give bit-identical factors.  Implies no zone
 * pipeline and no early slices.
 * Bit 13 / bit 14 = (multi-GPU) top tree distributed over the ranks / replicated on every rank
 * (default: by the weight of the top tree, see spllt_hip_set_partition).
 * Bit 15 / bit 16 = HIP-graph replay of the factorization (analyse once, factorize many): one
 * graph per pattern built from the program tables, a chain of kernel nodes in program order / the
 * DAG of the multi-stream program; bit 17 = eager launches.  Default: by problem size -- up to
 * 40 GFLOP the chain replay, and for it the SINGLE-STREAM program (a chain runs in program order
 * anyway: no zones, slices, markers or events -- 27 instead of 36 kernels on BASELINE config 1):
 * 0.22 vs 0.48 ms eager at the smoke size, 0.36 vs 0.69 ms on BASELINE config 1, 2.80 vs 3.28 ms at
 * 19 GFLOP, 4.02 vs 4.66 ms at 33 GFLOP; eager launches above (level at 313 GFLOP: on ROCm 7.2
 * hipGraphLaunch submits nothing before all nodes are enqueued; 25.1 / 24.4 ms against 23.7 eager
 * on the 650 launches of the bench workload).  SPLLT_CHAIN_GRAPH_SERIAL=0 keeps the multi-stream
 * program under the chain replay.
 * Bit 18 / bit 19 = small subtrees as single device tasks on / off (L_SUBTREE, k_subtree: one
 * workgroup factorizes a whole subtree of one-panel nodes in post-order, what leaves the subtree
 * goes through a private generated element and reaches the ancestors in ONE extend-add from the
 * subtree's root -- the reference's pruned-subtree task, src/spllt_factorization_mod.F90:39-261).
 * Default off: slower than the level-batched launches at every size and budget measured
 * (profiles/r04/subtree_tasks_ab.txt); env SPLLT_SUBTREES=1, SPLLT_SUBTREE_US=<modelled us per task>.
 * Single-GPU, non-deterministic engines only.
 * Bits 2-5 selected round-1 experiments that have been removed.
 * Every variant produces the same factor (tests/test_gpu_parity.py). */
int spllt_hip_set_engine(void *fkeep, int panel_width, int tile, int flags);
/* Accepted and ignored.  (It set the edge of diagonal sub-tiles that a single-workgroup chain
 * kernel walked, several panels per sub-tile; that variant was slower at every setting and has
 * been removed: the chain block is always one panel.) */
int spllt_hip_set_chain_block(void *fkeep, int chain_block);

/* ---- multi-GPU: one process per GPU, subtree partition ----------------------
 * Call after spllt_analyse (options.prune_tree = 1, options.ncpu = nranks) and
 * before the first spllt_factor.  Every rank factorizes the pruned subtrees it
 * owns; spllt_factor then stops at the first EXCHANGE point of the rank's program.  The
 * caller drives the exchanges:
 *
 *   spllt_hip_factor_dev(...);
 *   while ((k = spllt_hip_pending_exchange(fkeep)) >= 0) {
 *     <collective of exchange k on the exchange buffer, enqueued on spllt_hip_engine_stream>
 *     spllt_hip_continue(fkeep);        // unpack, enqueue up to the next exchange, pack
 *   }
 *   spllt_hip_wait(fkeep);
 *
 * Exchange k is described by spllt_hip_program_get "exchanges" (int64 x 5: kind, first item,
 * items, elems, chunk) and "xitems" (int64 x 6: block column, root, offset in the buffer, count,
 * offset in the arena or the dinv scratch, space 0 = arena / 1 = dinv scratch); every rank
 * has the same list.  kind 0: all-reduce(sum) of buffer[0:elems] - the extend-add of the whole
 * top tree (replaces spllt_scatter_block on generated elements, reference
 * src/spllt_factorization_mod.F90:39-191; the last element carries the "not positive definite"
 * indicator); the top tree is then factorized on every rank (replicated).
 * Distributed top tree (engine flag bit 13, or chosen by the engine when the top tree is
 * heavy): kind 1: reduce-scatter(sum) of the REGION buffer[base:elems], base = elems - nranks *
 * chunk, in nranks chunks, rank r receives chunk r AT buffer[base + r*chunk : base + (r+1)*chunk]
 * (the block columns of ONE LEVEL of the top tree that r owns: there is one such exchange per
 * level of the top tree, lowest level first, each with a region of its own, so that the lowest
 * level is factorized while the chunks of the levels above travel; spllt_hip_exchange_stream);
 * kind 2:
 * for every root with items, broadcast of that root's segment of the buffer (its items are
 * contiguous) - the block columns of a finished step go from their owners to all ranks, which
 * then update the destination block columns they own; kind 3: all-reduce(sum) of buffer[0:1],
 * the "not positive definite" indicator, at the very end.
 * exchange_elems: doubles the exchange buffer must hold (all reduce regions + the largest broadcast). */
int spllt_hip_set_partition(void *fkeep, int rank, int nranks, int64_t *exchange_elems);
/* The collectives INSIDE the library: hand over the caller's RCCL communicator (ncclComm_t, one
 * rank per GPU, its rank / size = the partition's) after spllt_hip_set_partition.  From then on
 * the unchanged spllt_iface.h calls are all a C or Fortran caller needs: spllt_factor enqueues
 * the rank's subtrees, every exchange of the program (all-reduce of the top tree, or
 * reduce-scatter to the owners + one broadcast per block-column step, and the flag) between its
 * pack and unpack on the engine's stream, and the top tree; spllt_wait drains it; spllt_solve
 * (job 0) adds the two all-reduces of the right-hand sides.  The exchange buffer is then the
 * library's own.  What the reference's distributed build keeps inside the library too
 * (src/PaRSEC/spllt_parsec_blk_data.c:33-64, factorize.jdf).  librccl is resolved at run time
 * from the copy the process already uses.  NULL detaches (the caller drives the exchanges again,
 * spllt_hip_pending_exchange / spllt_hip_continue). */
int spllt_hip_set_communicator(void *fkeep, void *nccl_comm);
/* index of the exchange the handle is waiting for (-1: none, the program has been enqueued
 * to its end) */
int spllt_hip_pending_exchange(void *fkeep);

/* Substitution on DEVICE vectors in pivot order: y_dev holds nrhs vectors of
 * length n, y[q*n + p(i)] = b_q[i], p = 0-based pivot position ("order" of spllt_hip_sym_get); overwritten by the solution in the same
 * order.  job as spllt_solve (0 both sweeps, 1 forward, 2 backward; reference
 * src/spllt_solve_mod.F90:203-221).  phase = -1: the whole solve (single GPU).
 * On a partitioned factor the solve runs in three phases with the caller's
 * exchange, like the factorization:
 *   y = b on the entries this rank owns (own subtrees; rank 0 also the top tree), 0 elsewhere
 *   phase 0 (forward, own subtrees)            -> all-reduce(sum) of y over the ranks
 *   phase 1 (top tree, forward and backward), phase 2 (backward, own subtrees)
 *   zero the entries this rank does not own    -> all-reduce(sum): x on every rank.
 * spllt_solve itself returns SPLLT_ERROR_UNIMPLEMENTED on a partitioned factor. */
int spllt_hip_solve_dev(void *fkeep, void *y_dev, int nrhs, int job, int phase);
int spllt_hip_set_exchange_buffer(void *fkeep, void *dev_ptr);
/* The HIP stream (hipStream_t) every caller-visible operation of this handle is ordered on:
 * spllt_factor ends by packing the exchange buffer on it and spllt_hip_continue starts by
 * unpacking it there.  A caller that enqueues its collective ON this stream (RCCL takes a
 * stream argument; torch: torch.cuda.ExternalStream) needs no host synchronisation between
 * the phases of a partitioned factorization.  Creates the device engine if necessary;
 * NULL without a HIP device. */
void *spllt_hip_engine_stream(void *fkeep);
/* The stream the PENDING exchange (spllt_hip_pending_exchange) is packed and unpacked on: the
 * stream of spllt_hip_engine_stream, except for the per-level reduce-scatters of a distributed
 * top tree, which the multi-stream program issues on a side stream so that the chunks of the
 * upper top-tree levels travel while the lowest one is already being factorized (SURVEY 8(e):
 * "pipeline per ancestor node").  The caller's collective for that exchange goes on THIS stream. */
void *spllt_hip_exchange_stream(void *fkeep);
int spllt_hip_continue(void *fkeep);
/* "arena_elems" (int64 x 2: doubles of the factor arena held on this rank's device once its engine
 * exists - a rank stores only its own branches and the top tree, packed -, doubles of the whole arena),
 * "owner" (int32 per node: rank or -1 = top tree), "top_bcols" (int32),
 * "top_bcol_owner" (int32 per block column: owner in a distributed top tree, -1 outside it;
 * empty when the top tree is replicated),
 * "map_keep" (uint8 per val->L map entry: scattered on this rank) */
int64_t spllt_hip_partition_get(void *fkeep, const char *name, void *buf, int64_t capacity_bytes);

/* spllt_factor with val already resident in HBM (device pointer).  val_dev must stay valid and
 * unchanged until spllt_hip_wait returns: the factorization reads it in place. */
void spllt_hip_factor_dev(void *akeep, void *fkeep, spllt_options_t *options, int nnz,
                          const double *val_dev, spllt_inform_t *info);
/* wait for ONE factorization and return its flag (spllt_wait() has no way to) */
int spllt_hip_wait(void *fkeep);
/* copy the whole L arena (block columns concatenated in order) to host memory */
int spllt_hip_get_factor(void *fkeep, double *out, int64_t count);
/* device pointer of the L arena (valid until spllt_deallocate_fkeep) */
/* (partitioned factorization: the rank's packed arena, not the global layout) */
double *spllt_hip_device_factor(void *fkeep);
/* timings of the last factorization, milliseconds */
int spllt_hip_factor_times(void *fkeep, double *submit_ms, double *device_ms, double *h2d_ms,
                           int *launches);
/* program export for tests: "launches" (int64 x 12 per launch: kind, level,
 * first, count, tile, flops, stream, record, wait0..wait3), "chains" (ChainUnit bytes),
 * "potrf" (PotrfUnit bytes), "units" (UpdUnit bytes), "tiles" (UpdTile bytes),
 * "relpos" (int32), "dinv_size" (int64), "chain_block" (int64), "gather_tiles" / "gather_items"
 * (GatherTile / GatherItem bytes), "scratch_size" (int64); the substitution program:
 * "solve_units" (SolveUnit bytes), "solve_list" (int32), "solve_tiles" (UpdTile bytes),
 * "solve_fwd" / "solve_bwd" (int64 x 4 per launch: kind, level, first, count), "solve_split"
 * (int64 x 2: launches of fwd that belong to the own branches, launches of bwd that belong
 * to the top tree).  Struct layouts: spllt_amd/csrc/schedule.hpp, mirrored as numpy dtypes in
 * spllt_amd/api.py.  Returns the byte length. */
int64_t spllt_hip_program_get(void *fkeep, const char *name, void *buf, int64_t capacity_bytes);
/* per-launch device time (ms) of one profiled factorization; returns #launches */
int spllt_hip_profile(void *fkeep, const double *val, int nnz, float *ms, int capacity);
/* the same for the real multi-stream program: every launch bracketed by two HIP events on
 * the stream it runs on (behind its dependency waits), i.e. its duration beside whatever the
 * other streams run at that moment */
int spllt_hip_profile_in_program(void *fkeep, const double *val, int nnz, float *ms, int capacity);
/* when every event of the real multi-stream program was reached: t_ms[i] = ms after the value
 * scatter at which the event that launch i records completed (-1: the launch records none),
 * t_ms[#launches] = the end of the program.  Nothing is added to the streams (the program's own
 * records, on timing-enabled events).  Returns #launches + 1. */
int spllt_hip_timeline(void *fkeep, const double *val, int nnz, float *t_ms, int capacity);
const char *spllt_hip_last_error(const void *fkeep);
/* flag of the last operation on this handle (0, or an SPLLT error flag): what spllt_wait(void),
 * which has no way to return it, found when the factorization had run */
int spllt_hip_last_flag(const void *fkeep);
const char *spllt_hip_version(void);

/* test hooks: "wedge" marks the HIP runtime as not having returned from a call (what the wait /
 * submission deadlines do), "wedged" reads the mark (1 / 0), "teardown" runs the library's atexit
 * handler now (it must touch nothing once the mark is set).  -1: unknown request. */
int spllt_hip_debug(const char *what);

#ifdef __cplusplus
}
#endif
#endif
