// Device-side factorization engine: owns the L arena in HBM, the uploaded
// work tables of the stream-DAG program, and the HIP stream(s) it runs on.
// This is what replaces spllt_stf_factorize + the task runtimes
// (reference src/spllt_stf_mod.F90:18-192).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "schedule.hpp"
#include "symbolic.hpp"

namespace spx {

// deadline of every blocking wait, seconds (SPLLT_HIP_TIMEOUT_S, default 180; 0: none)
double hip_deadline_s();
// runs fn on the process's submission thread and waits for it with that deadline; a job that
// does not come back makes this call and every later one return SPLLT_ERROR_HIP (-30) with *why
// naming the last step the library reached (engine.cpp)
int run_with_deadline(std::function<int()> fn, std::string* why);
// process-wide "the HIP runtime did not come back from a call" flag (set by the wait / submission
// deadlines): the atexit teardown of the pools then touches nothing
void mark_runtime_wedged();
bool runtime_wedged();
void run_pools_teardown_for_test();
const char* last_crumb();

struct EngineOptions {
  int pw = 64;
  int tile = 128;
  int cb = 64;             // ignored (see ScheduleOptions)
  bool lookahead = true;   // multi-stream program (panel chain overlaps trailing updates)
  bool slice_between = true;  // inter-node updates in K slices beside the panel chains
  bool deterministic = false;  // see ScheduleOptions
  bool fused_panel = true;     // see ScheduleOptions
  int dist_top = -1;           // multi-GPU top tree: 1 distributed over the ranks, 0 replicated on every
                               // rank, -1 by distribute_top_tree() (env SPLLT_DIST_TOP overrides)
  bool poison_lds = false; // debug: poison the LDS of every CU before every launch
  int reserve_cus = -1;    // CUs the bulk / far streams are masked off (0: no mask; -1: 32 when the
                           // problem is latency-bound (schedule.hpp), else 0)
  int zones = -1;          // zone pipeline of the inter-node updates (1 / 0; -1: when latency-bound)
  int rank = 0, nranks = 1;  // multi-GPU subtree partition (nranks > 1: two-phase program)
  int subtrees = -1;       // small subtrees as single device tasks (L_SUBTREE): 1 / 0; -1: the default (off)
  int graph = -1;          // HIP-graph replay of the factorization (single GPU): 0 eager launches, 1 one
                           // chain of kernel nodes in program order, 2 the DAG of the multi-stream
                           // program, -1 by problem size (env SPLLT_HIP_GRAPH overrides)
};

struct FactorStats {
  double submit_ms = 0;   // host time spent in factor_async
  double device_ms = 0;   // HIP-event time from first to last enqueued operation
  double h2d_ms = 0;
  int launches = 0;
};

// EngineOptions -> ScheduleOptions (the one place); resolves the "decide by the problem" options of opt
ScheduleOptions schedule_options(const Symbolic& S, EngineOptions& opt);
// HIP-graph replay of a factorization of S: 0 eager launches, 1 one chain of kernel nodes, 2 the DAG of the
// multi-stream program (opt.graph, env SPLLT_HIP_GRAPH, or by problem size)
int resolve_graph_mode(const Symbolic& S, const EngineOptions& opt);

// Partition of the tree for opt.nranks ranks: node owners, and (distributed top tree) the owners of
// the top-tree block columns; fills the partition fields of so (the vectors must outlive it).
void partition_options(const Symbolic& S, const EngineOptions& opt, std::vector<int>& owner,
                       std::vector<int>& top_owner, ScheduleOptions& so);

class Engine {
 public:
  Engine(std::shared_ptr<const Symbolic> S, const EngineOptions& opt);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  int status() const { return status_; }  // 0 or SPLLT error flag from construction
  bool poisoned() const { return poisoned_; }
  bool comm_rehearsal() const { return comm_rehearsal_; }
  void poison(const std::string& why) { poisoned_ = true; status_ = -30; err_ = why; }
  const std::string& error() const { return err_; }

  // spllt_factor: enqueue H2D of val, value scatter and the whole program.
  int factor_async(const double* val_host, int64_t nnz);
  // Same with val already resident in HBM (device pointer, same device).
  int factor_async_dev(const double* val_dev, int64_t nnz);
  // spllt_wait for this engine: drain the stream, surface "not positive definite".
  int wait();
  bool pending() const { return pending_; }
  // ---- multi-GPU (nranks > 1): factor_async* stops after the rank's own
  // subtrees with the top-tree block columns packed into the exchange buffer;
  // the caller reduces that buffer across ranks (RCCL all-reduce), then calls
  // continue_after_exchange() and finally wait().
  // doubles the exchange buffer must hold (the largest exchange of the program)
  int64_t exchange_elems() const { return prog_.xbuf_elems; }
  int set_exchange_buffer(double* dev_ptr) { xbuf_ = dev_ptr; return 0; }
  bool awaiting_exchange() const { return awaiting_exchange_; }
  // index (into program().exchanges) of the exchange the engine is waiting for, -1: none
  int pending_exchange() const { return awaiting_exchange_ ? (int)prog_.launches[cur_x_].first : -1; }
  // ---- the collectives inside the library (RCCL over xGMI): with a communicator set,
  // factor_async* does not stop at the exchange points -- every exchange of the program
  // (all-reduce of the top tree / reduce-scatter to the owners + one broadcast per block-column
  // step / the flag) is enqueued on the engine's stream between its pack and its unpack, and
  // the solve adds its two all-reduces of the right-hand sides.  What the reference's
  // distributed build has inside the library too (src/PaRSEC/spllt_parsec_blk_data.c:33-64,
  // factorize.jdf).  comm: the caller's ncclComm_t, one rank per GPU.
  int set_communicator(void* nccl_comm);
  bool has_communicator() const { return comm_ != nullptr; }
  int run_exchanges();              // all pending exchanges, enqueue only
  int sync_phase();                 // drain the streams at the exchange point
  int continue_after_exchange();
  const std::vector<int>& owners() const { return owner_; }
  const std::vector<int>& top_bcols() const { return top_bcols_; }
  const std::vector<char>& map_keep() const { return map_keep_; }
  int not_posdef_column() const { return npd_col_; }
  // doubles of the factor arena held on this device (a rank of a partition: own branches + top tree)
  int64_t arena_elems() const { return arena_elems_; }

  int download(double* out, int64_t count);  // D2H of the arena
  // spllt_solve on the device-resident factor (x: n x nrhs column-major, original
  // variable order, overwritten).  job 0 = both sweeps, 1 = forward, 2 = backward.
  int solve(double* x_host, int nrhs, int job);
  // substitution on device vectors in pivot order; phase -1 = all, 0/1/2 = partitioned phases
  int solve_dev(double* y_dev, int nrhs, int job, int phase);
  int prepare_solve();
  double* device_L() { return d_L_; }
  hipStream_t stream() { return stream_; }
  // the stream the pending exchange is packed / unpacked on (its collective belongs there); the chain stream when none is pending
  hipStream_t pending_exchange_stream() const { return awaiting_exchange_ ? exchange_stream(prog_.launches[cur_x_]) : stream_; }
  const Program& program() const { return prog_; }
  const Symbolic& symbolic() const { return *S_; }
  const FactorStats& stats() const { return stats_; }
  // per-launch device time of the last factorization (profiling mode)
  int profile_launches(const double* val_host, int64_t nnz, std::vector<float>& ms, bool serial = true);
  // when each event of the real program was reached (ms after the value scatter; see engine.cpp)
  int timeline(const double* val_host, int64_t nnz, std::vector<float>& t);

 private:
  int upload();
  int enqueue_program();
  int enqueue_range(size_t first, size_t last);
  int run_from(size_t first);                 // enqueue launches until the next exchange or the end
  int pre_exchange(const Launch& X);          // waits + pack
  int post_exchange(const Launch& X);         // unpack + record
  int finish_enqueue();
  int enqueue_launch(const Launch& l, bool serial);
  void emit_kernel(const Launch& l, const struct LaunchSink& sink, bool multi);
  int build_graph(int mode);
  int fail(int code, const char* what, hipError_t e);
  // hipStreamSynchronize with a deadline (SPLLT_HIP_TIMEOUT_S, default 180 s; 0 = wait forever):
  // a stream that does not drain makes the call FAIL with a report of the first launch of the
  // program whose event has not fired, instead of blocking the caller forever
  int sync_stream(hipStream_t st, const char* what);

  std::shared_ptr<const Symbolic> S_;
  EngineOptions opt_;
  Program prog_;
  int status_ = 0;
  std::string err_;
  int device_ = 0;
  // streams of the program (schedule.hpp StreamId).  stream_ = chain stream, also the
  // stream every caller-visible operation (H2D, pack, solve) is ordered on.
  hipStream_t stream_ = nullptr;
  hipStream_t streams_[ST_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int chain_prio_ = 1;                 // s_setprio for chain / side update launches
  // optional dynamic-LDS padding (bytes) of the trailing updates that run beside a panel
  // chain (caps their workgroups per CU); off by default: the bulk streams are masked off
  // reserve_cus CUs instead
  int bulk_pad128_ = 0, bulk_pad64_ = 0;
  std::vector<hipEvent_t> dag_events_;  // dependency events of the program
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr, ev_h2d_ = nullptr, ev_init_ = nullptr;
  // device buffers of this engine (pointer, bytes): taken from / returned to the process-wide cache
  std::vector<std::pair<void*, size_t>> owned_;
  hipError_t dalloc(void** p, size_t bytes);
  template <class Tp> hipError_t dev_upload(Tp** dptr, const std::vector<Tp>& v);
  // host-to-device copy of val through two pinned staging buffers (spllt_factor's val is pageable
  // user memory: handing it to hipMemcpyAsync makes the "asynchronous" copy a blocking one inside
  // the runtime, which pins or stages it there)
  static constexpr size_t kH2dChunk = (size_t)16 << 20, kH2dSmall = (size_t)256 << 10;
  size_t h2d_chunk_ = 0;
  void* h2d_buf_[2] = {nullptr, nullptr};
  hipEvent_t h2d_ev_[2] = {nullptr, nullptr};
  bool h2d_busy_[2] = {false, false};
  int stage_val(const double* val_host, int64_t nnz);
  int staged_d2h(void* out_host, const void* src_dev, size_t bytes);
  int wait_event(hipEvent_t ev, const char* what);
  bool poisoned_ = false;   // a wait ran into its deadline: see ~Engine
  bool comm_rehearsal_ = false;   // set_communicator accepted a one-rank stand-in (SPLLT_HIP_COMM_REHEARSAL): results are not the factor
  mutable bool localize_failed_ = false;
  int graph_mode_ = 0;
  const double* val_src_ = nullptr;   // eager factor_async_dev: the caller's device array, read in place (else d_val_)
  bool replays_graph() const { return graph_mode_ > 0 && prog_.exchanges.empty() && !opt_.poison_lds; }
  hipGraph_t graph_ = nullptr;
  hipGraphExec_t graph_exec_ = nullptr;
  bool pending_ = false;
  bool awaiting_exchange_ = false;
  size_t cur_x_ = 0;                // launch index of the exchange the engine waits for
  std::vector<int> top_owner_;      // per block column: owner in a distributed top tree (else empty)
  double* xbuf_ = nullptr;          // caller-owned device buffer of xchg_elems_ doubles
  void* comm_ = nullptr;            // ncclComm_t (set_communicator)
  int comm_rank_ = 0, comm_size_ = 1;
  int collective(const Exchange& E, hipStream_t xs);
  hipStream_t exchange_stream(const Launch& X) const;
  double* d_owned_ = nullptr;       // per pivot position: 1.0 where this rank contributes to a distributed vector
  std::vector<int> owner_;          // per node: owning rank or -1 (top tree)
  std::vector<int> top_bcols_;      // block columns of the top tree, in order
  std::vector<char> map_keep_;      // per val->L map entry: scattered on this rank?
  std::vector<std::pair<int64_t, int64_t>> zero_ranges_;  // (offset, count) of the arena this rank clears
  int64_t nmap_ = 0;                // entries of the (filtered) scatter map on the device
  int npd_col_ = -1;
  // A rank of a partition stores only the block columns it works on (its own branches + the top
  // tree), packed: loc_off_[b] = offset of block column b in THIS rank's arena (-1: not held
  // here).  Every table that carries an arena offset is translated after the program is built;
  // the host-side view (download) stays in the global layout.  Empty: one GPU, arena = global.
  std::vector<int64_t> loc_off_;
  int64_t arena_elems_ = 0;          // doubles of the device arena
  int64_t to_local(int64_t global_off) const;
  void localize_program();
  FactorStats stats_;

  double* d_L_ = nullptr;
  double* d_val_ = nullptr;
  double* d_dinv_ = nullptr;
  int64_t* d_map_dst_ = nullptr;
  int64_t* d_init_cptr_ = nullptr;          // val -> L map bucketed by arena chunk (k_init_arena; single GPU)
  unsigned short* d_init_loc_ = nullptr;
  int* d_init_src_ = nullptr;
  int64_t* d_map_src_ = nullptr;
  int64_t* d_bc_off_ = nullptr;
  int* d_bc_w_ = nullptr;
  char* d_tables_ = nullptr;        // one allocation behind the table pointers below
  char* d_solve_tables_ = nullptr;  // ... and behind the solve tables
  UpdUnit* d_units_ = nullptr;
  UpdTile* d_tiles_ = nullptr;
  ChainUnit* d_chain_ = nullptr;
  PanelUnit* d_panel_ = nullptr;
  SubTask* d_sub_tasks_ = nullptr;   // L_SUBTREE: one workgroup per small subtree
  SubNode* d_sub_nodes_ = nullptr;
  double* d_gen_ = nullptr;          // generated elements of the subtree tasks (zero between factorizations)
  bool solve_four_ = false;        // the solve may use k_solve_diag4 (panels of 64)
  int* d_panel_cnt_ = nullptr;     // two "last reader" counters per panel unit (zero between launches)
  GatherTile* d_gtiles_ = nullptr;
  GatherItem* d_gitems_ = nullptr;
  double* d_scratch_ = nullptr;    // MODE_BUFFER products (deterministic engine)
  // device solve (built on first use)
  SolveProgram sprog_;
  bool solve_ready_ = false;
  SolveUnit* d_sunits_ = nullptr;
  int* d_slist_ = nullptr;
  UpdTile* d_stiles_ = nullptr;
  double* d_y_ = nullptr;
  int* d_relpos_ = nullptr;
  int* d_rlist_ = nullptr;
  int* d_flag_ = nullptr;
  int* h_flag_ = nullptr;  // pinned
};

}  // namespace spx
