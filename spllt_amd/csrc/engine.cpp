// Engine implementation (HIP runtime API; compiled by hipcc as host code).
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <dlfcn.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace spx {

namespace {
constexpr int kErrHip = -30, kErrNotPosDef = -20;
double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
}  // namespace

#define HIPCHK(call, what)                                   \
  do {                                                       \
    hipError_t e__ = (call);                                 \
    if (e__ != hipSuccess) return fail(kErrHip, what, e__);  \
  } while (0)

int Engine::fail(int code, const char* what, hipError_t e) {
  status_ = code;
  err_ = std::string(what) + ": " + hipGetErrorString(e);
  std::fprintf(stderr, "spllt-hip: %s\n", err_.c_str());
  return code;
}

// Tables of an engine go to the device as ONE allocation and ONE copy: the parts are laid out in
// a host staging buffer (256-byte aligned), the typed device pointers are set after the upload.
// (Two dozen allocations, synchronous copies and frees per engine were two dozen chances per
// engine to sit in the runtime.)
struct TableStager {
  struct Slot { void** dptr; size_t off; };
  std::vector<char> host;
  std::vector<Slot> slots;
  template <class Tp>
  void add(Tp** dptr, const Tp* src, size_t count) {
    const size_t off = (host.size() + 255) / 256 * 256;
    const size_t bytes = std::max<size_t>(count * sizeof(Tp), 8);
    host.resize(off + bytes, 0);
    if (count) std::memcpy(host.data() + off, src, count * sizeof(Tp));
    slots.push_back({(void**)dptr, off});
  }
  template <class Tp>
  void add(Tp** dptr, const std::vector<Tp>& v) { add(dptr, v.data(), v.size()); }
  template <class Alloc>
  hipError_t commit(char** blob, Alloc&& alloc) {
    hipError_t e = alloc((void**)blob, std::max<size_t>(host.size(), 8));
    if (e != hipSuccess) return e;
    if (!host.empty()) e = hipMemcpy(*blob, host.data(), host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    for (const Slot& sl : slots) *sl.dptr = *blob + sl.off;
    return hipSuccess;
  }
};

// pinned words for the "not positive definite" flag, pooled per process
static std::vector<int*> g_pinned_words;
static std::mutex g_pinned_mu;
static hipError_t borrow_pinned_word(int** p) {
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  if (!g_pinned_words.empty()) {
    *p = g_pinned_words.back();
    g_pinned_words.pop_back();
    return hipSuccess;
  }
  return hipHostMalloc((void**)p, sizeof(int), hipHostMallocDefault);
}
static void return_pinned_word(int* p) {
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  if (p) g_pinned_words.push_back(p);
}

// ---------------------------------------------------------------------------
// Process-wide resource pools.  An engine takes its runtime objects from here and hands them
// back; nothing is created or destroyed per engine that can be reused:
//   * streams and events (below): creating and destroying a CU-masked stream is creating and
//     destroying a hardware queue (the driver unmaps and remaps every queue of the process for
//     it) -- the one thing the engines of round 2 did per factorization that nothing else in
//     the process does, see DESIGN.md "Hangs";
//   * device buffers: hipFree synchronises the WHOLE device -- every other live engine's
//     streams, and the caller's -- so the buffers of a closed engine are kept (up to a cap) for
//     the next one;
//   * pinned staging buffers for the host-to-device copy of val.
// pools_teardown() (atexit, registered on first use, i.e. after the HIP runtime has registered
// its own handlers and therefore run BEFORE them) drains and destroys all of it, so that the
// runtime does not unload with live CU-masked queues (the exit crash of the profiled runs).
// ---------------------------------------------------------------------------
namespace {
std::mutex g_pool_mu;
struct DevBuf { void* p; size_t bytes; int device; };
std::vector<DevBuf> g_dev_cache;
size_t g_dev_cached_bytes = 0;
struct PinBuf { void* p; size_t bytes; };
std::vector<PinBuf> g_pin_cache;
bool g_teardown_registered = false;
void pools_teardown();
// Process-wide: a wait ran into its deadline or a submission never returned.  From then on the
// HIP runtime may be stuck inside a call that holds its own locks (or ours): nothing at exit may
// touch it again -- pools_teardown() returns at once, so that the process that DETECTED the hang
// can still exit (its exit code tells the caller), instead of trading "the call never returns"
// for "the process never exits".
std::atomic<bool> g_runtime_wedged{false};
void register_teardown() {
  if (!g_teardown_registered) {
    g_teardown_registered = true;
    std::atexit(pools_teardown);
  }
}
size_t dev_cache_cap() {
  static const size_t cap = [] {
    const char* e = std::getenv("SPLLT_HIP_CACHE_MB");
    return (size_t)(e && *e ? std::atoll(e) : 2048) << 20;
  }();
  return cap;
}
}  // namespace

// a device buffer of at least `bytes`: from the cache (smallest fit that wastes at most half), else
// hipMalloc; *cap receives the buffer's real capacity (what dev_release must be told, so that the
// cache's byte count -- and with it the SPLLT_HIP_CACHE_MB cap -- stays exact)
static hipError_t dev_alloc(void** p, size_t bytes, int device, size_t* cap) {
  bytes = std::max<size_t>(bytes, 256);
  *cap = bytes;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    register_teardown();
    int best = -1;
    for (size_t i = 0; i < g_dev_cache.size(); ++i) {
      const DevBuf& b = g_dev_cache[i];
      if (b.device != device || b.bytes < bytes || b.bytes > 2 * bytes + (1 << 20)) continue;
      if (best < 0 || b.bytes < g_dev_cache[(size_t)best].bytes) best = (int)i;
    }
    if (best >= 0) {
      *p = g_dev_cache[(size_t)best].p;
      *cap = g_dev_cache[(size_t)best].bytes;
      g_dev_cached_bytes -= g_dev_cache[(size_t)best].bytes;
      g_dev_cache.erase(g_dev_cache.begin() + best);
      return hipSuccess;
    }
  }
  return hipMalloc(p, bytes);
}
// back into the cache (the caller has drained every stream that used it); beyond the cap, or
// when the size is not known to the cache, it is freed for real
static void dev_release(void* p, size_t bytes, int device) {
  if (!p) return;
  bytes = std::max<size_t>(bytes, 256);
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_dev_cached_bytes + bytes <= dev_cache_cap()) {
      g_dev_cache.push_back({p, bytes, device});
      g_dev_cached_bytes += bytes;
      return;
    }
  }
  (void)hipFree(p);
}
static hipError_t pin_alloc(void** p, size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    register_teardown();
    for (size_t i = 0; i < g_pin_cache.size(); ++i)
      if (g_pin_cache[i].bytes == bytes) {
        *p = g_pin_cache[i].p;
        g_pin_cache.erase(g_pin_cache.begin() + (long)i);
        return hipSuccess;
      }
  }
  return hipHostMalloc(p, bytes, hipHostMallocDefault);
}
static void pin_release(void* p, size_t bytes) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pin_cache.push_back({p, bytes});
}

double hip_deadline_s() {
  static const double limit_s = [] {
    const char* e = std::getenv("SPLLT_HIP_TIMEOUT_S");
    return (e && *e) ? std::atof(e) : 180.0;
  }();
  return limit_s;
}

// hipEventSynchronize with the deadline of sync_stream
int Engine::wait_event(hipEvent_t ev, const char* what) {
  const double limit_s = hip_deadline_s();
  if (limit_s <= 0) {
    hipError_t e = hipEventSynchronize(ev);
    return e == hipSuccess ? 0 : fail(kErrHip, what, e);
  }
  const double t0 = now_ms();
  for (;;) {
    hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return fail(kErrHip, what, q);
    const double dt = now_ms() - t0;
    if (dt > limit_s * 1e3) break;
    if (dt > 5.0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else std::this_thread::yield();
  }
  status_ = kErrHip;
  err_ = std::string(what) + ": the copy engine did not finish within " + std::to_string((int)limit_s) + " s";
  poisoned_ = true;
  mark_runtime_wedged();
  std::fprintf(stderr, "spllt-hip: %s\n", err_.c_str());
  return kErrHip;
}

// val (pageable user memory) -> d_val_ through two pinned 16 MiB buffers: the host copies chunk
// k + 1 while the DMA of chunk k flies; every wait has the deadline.
int Engine::stage_val(const double* val_host, int64_t nnz) {
  const size_t bytes = sizeof(double) * (size_t)nnz;
  // (two sizes of staging buffers: pinning 2 x 16 MiB costs ~2 ms the first time, which a small
  // problem need not pay)
  const size_t chunk = bytes <= kH2dSmall ? kH2dSmall : kH2dChunk;
  if (h2d_chunk_ != chunk) {
    for (int i = 0; i < 2; ++i) {
      if (h2d_busy_[i]) { int rc = wait_event(h2d_ev_[i], "val H2D staging"); if (rc) return rc; h2d_busy_[i] = false; }
      if (h2d_buf_[i]) pin_release(h2d_buf_[i], h2d_chunk_);
      h2d_buf_[i] = nullptr;
    }
    h2d_chunk_ = chunk;
  }
  for (int i = 0; i < 2; ++i) {
    if (!h2d_buf_[i]) HIPCHK(pin_alloc(&h2d_buf_[i], chunk), "hipHostMalloc(staging)");
    if (!h2d_ev_[i]) HIPCHK(hipEventCreateWithFlags(&h2d_ev_[i], hipEventDisableTiming), "hipEventCreate");
  }
  const char* src = reinterpret_cast<const char*>(val_host);
  char* dst = reinterpret_cast<char*>(d_val_);
  int k = 0;
  for (size_t off = 0; off < bytes; off += chunk, ++k) {
    const int i = k & 1;
    const size_t len = std::min(chunk, bytes - off);
    if (h2d_busy_[i]) {               // the buffer's previous DMA (this call's or the last one's)
      int rc = wait_event(h2d_ev_[i], "val H2D staging");
      if (rc) return rc;
    }
    std::memcpy(h2d_buf_[i], src + off, len);
    HIPCHK(hipMemcpyAsync(dst + off, h2d_buf_[i], len, hipMemcpyHostToDevice, stream_), "val H2D");
    HIPCHK(hipEventRecord(h2d_ev_[i], stream_), "event");
    h2d_busy_[i] = true;
  }
  return 0;
}

int Engine::sync_stream(hipStream_t st, const char* what) {
  const double limit_s = hip_deadline_s();
  if (limit_s <= 0) {
    hipError_t e = hipStreamSynchronize(st);
    return e == hipSuccess ? 0 : fail(kErrHip, what, e);
  }
  const double t0 = now_ms();
  for (;;) {
    hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return fail(kErrHip, what, q);
    const double dt = now_ms() - t0;
    if (dt > limit_s * 1e3) break;
    if (dt > 20.0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else std::this_thread::yield();
  }
  // the device does not finish: say where the program stands
  std::string rep = std::string(what) + ": the stream did not drain within " + std::to_string((int)limit_s) + " s;";
  int shown = 0;
  for (size_t i = 0; i < prog_.launches.size() && shown < 4; ++i) {
    const Launch& l = prog_.launches[i];
    if (l.record < 0 || hipEventQuery(dag_events_[(size_t)l.record]) == hipSuccess) continue;
    rep += " launch " + std::to_string(i) + " (kind " + std::to_string(l.kind) + ", level " + std::to_string(l.level) +
           ", count " + std::to_string((long long)l.count) + ", tile " + std::to_string(l.tile) + ", stream " +
           std::to_string(l.stream) + ") has not finished;";
    ++shown;
  }
  for (int i = 0; i < ST_COUNT; ++i)
    if (streams_[i]) rep += " stream " + std::to_string(i) + (hipStreamQuery(streams_[i]) == hipSuccess ? " idle;" : " busy;");
  status_ = kErrHip;
  err_ = rep;
  // The device may still be working on (or stuck in) what this engine enqueued: nothing of it is
  // touched again -- no further synchronisation (it would block for good), and its streams,
  // events, pinned words and device buffers are neither reused nor freed (see ~Engine).
  poisoned_ = true;
  mark_runtime_wedged();
  std::fprintf(stderr, "spllt-hip: %s\n", err_.c_str());
  return kErrHip;
}

hipError_t Engine::dalloc(void** p, size_t bytes) {
  size_t cap = 0;
  hipError_t e = dev_alloc(p, bytes, device_, &cap);
  if (e == hipSuccess) owned_.push_back({*p, cap});
  return e;
}

template <class Tp>
hipError_t Engine::dev_upload(Tp** dptr, const std::vector<Tp>& v) {
  size_t bytes = sizeof(Tp) * (v.empty() ? 1 : v.size());
  hipError_t e = dalloc((void**)dptr, bytes);
  if (e != hipSuccess) return e;
  if (!v.empty()) e = hipMemcpy(*dptr, v.data(), sizeof(Tp) * v.size(), hipMemcpyHostToDevice);
  return e;
}

// EngineOptions -> ScheduleOptions, in ONE place (the engine and the host-only program of
// spllt_hip_program_get / the CPU tests must build the same program); resolves the "-1 = decide by
// the problem" options of opt in place.
int resolve_graph_mode(const Symbolic& S, const EngineOptions& opt) {
  int mode = opt.graph;
  if (const char* e = std::getenv("SPLLT_HIP_GRAPH")) mode = std::atoi(e);
  if (mode < 0) {
    // by problem size (profiles/r04/graph_replay_by_size.txt): a small factorization is a few dozen
    // launches whose submission takes the host as long as the device needs for them -- the replay of
    // ONE chain of kernel nodes over the single-stream program wins 45-50 % at the small end (0.22 vs
    // 0.48 ms at the smoke size, 0.36 vs 0.69 ms on BASELINE config 1), 15 % at 19-33 GFLOP (2.80 /
    // 4.02 ms against 3.28 / 4.66 eager and 3.04 / 4.38 as the DAG replay), and is level with eager
    // launches at 313 GFLOP (hipGraphLaunch submits nothing before all nodes are enqueued)
    const double fl = (double)S.flops;
    mode = fl <= 40e9 ? 1 : 0;
  }
  if (opt.nranks > 1 || opt.poison_lds) mode = 0;      // (single-GPU programs without the debug poison only)
  return mode;
}

ScheduleOptions schedule_options(const Symbolic& S, EngineOptions& opt) {
  ScheduleOptions so;
  so.pw = opt.pw;
  so.tile = opt.tile;
  so.cb = opt.cb;
  so.lookahead = opt.lookahead;
  // a factorization that is replayed as ONE chain of kernel nodes runs in program order anyway: it gets
  // the single-stream program -- no zones, no early slices, no markers: 27 instead of 36 kernels on
  // BASELINE config 1 (profiles/r04/graph_replay_by_size.txt)
  {
    const char* e = std::getenv("SPLLT_CHAIN_GRAPH_SERIAL");      // (0: the multi-stream program, replayed in program order)
    if (!(e && std::atoi(e) == 0) && resolve_graph_mode(S, opt) == 1) so.lookahead = false;
  }
  so.slice_between = opt.slice_between;
  so.deterministic = opt.deterministic;
  so.fused_panel = opt.fused_panel;
  if (opt.subtrees >= 0) so.subtrees = opt.subtrees != 0;
  const bool lb = latency_bound(S, std::min(opt.pw, kPanelMax));
  // CU reservation (bulk / far streams masked off the last CUs): OFF by default since round 4.  It
  // bought 24.9 -> 24.4 ms in round 2 and buys 23.35 -> 23.2 ms now (0.6 %, profiles/r04/ab_cu_reserve.txt),
  // and CU-masked streams are the one object of this library whose destruction can hang
  // (profiles/r03/hang_evidence.txt) and whose survival makes rocprofv3 crash at exit.  Opt in
  // with SPLLT_HIP_RESERVE_CUS=32.
  if (opt.reserve_cus < 0) opt.reserve_cus = 0;
  if (opt.zones < 0) opt.zones = lb ? 1 : 0;
  so.zones = opt.zones != 0;
  // throughput-bound problems use the (LDS-DMA) 128-tile from 1024 tiles on: +0.3-0.9 % on the
  // large configurations; the latency-bound bench workload prefers 4096 (24.5 vs 25.0 ms)
  so.tile128_min = lb ? 4096 : 1024;
  return so;
}

void partition_options(const Symbolic& S, const EngineOptions& opt, std::vector<int>& owner,
                       std::vector<int>& top_owner, ScheduleOptions& so) {
  owner.clear();
  top_owner.clear();
  if (opt.nranks <= 1) return;
  assign_owners(S, opt.nranks, owner);
  so.node_owner = owner.data();
  so.rank = opt.rank;
  so.nranks = opt.nranks;
  int dist = opt.dist_top;
  if (const char* e = std::getenv("SPLLT_DIST_TOP"))
    if (*e) dist = std::atoi(e);
  if (dist < 0) dist = distribute_top_tree(S, owner, opt.nranks) ? 1 : 0;
  if (dist) {
    assign_top_owners(S, owner, opt.nranks, top_owner);
    so.top_owner = top_owner.data();
  }
}

Engine::Engine(std::shared_ptr<const Symbolic> S, const EngineOptions& opt)
    : S_(std::move(S)), opt_(opt) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    // The product has no CPU path: fail loudly.
    status_ = kErrHip;
    err_ = "no HIP device available (the factorize path runs only on gfx950)";
    std::fprintf(stderr, "spllt-hip: %s\n", err_.c_str());
    return;
  }
  hipGetDevice(&device_);
  ScheduleOptions so = schedule_options(*S_, opt_);
  if (opt_.nranks > 1) {
    partition_options(*S_, opt_, owner_, top_owner_, so);
    for (int b = 0; b < S_->nbcol(); ++b)
      if (owner_[S_->bcols[b].node] < 0) top_bcols_.push_back(b);
    // only the block columns this rank touches (its own branches + the top tree) are cleared
    // per factorization: the others are never read or written here
    for (int b = 0; b < S_->nbcol(); ++b) {
      const int own = owner_[S_->bcols[b].node];
      if (own != opt_.rank && own >= 0) continue;
      const int64_t off = S_->bcols[b].off, cnt = (int64_t)S_->bcols[b].nrow * S_->bcols[b].width;
      if (!zero_ranges_.empty() && zero_ranges_.back().first + zero_ranges_.back().second == off)
        zero_ranges_.back().second += cnt;
      else
        zero_ranges_.push_back({off, cnt});
    }
  }
  build_program(*S_, so, prog_);
  arena_elems_ = S_->arena;
  if (opt_.nranks > 1) localize_program();
  if (status_ == 0) upload();
}

// Breadcrumbs (SPLLT_HIP_CRUMBS=<file>): the last step the library reached, rewritten at every
// step.  After a call that never returned (a wedged runtime call cannot be interrupted or traced
// from the caller) the file names the step.
static std::atomic<const char*> g_last_crumb{"(nothing yet)"};
const char* last_crumb() { return g_last_crumb.load(); }
static void crumb(const char* what) {
  g_last_crumb.store(what);
  static const char* path = std::getenv("SPLLT_HIP_CRUMBS");
  if (!path || !*path) return;
  if (FILE* f = std::fopen(path, "w")) {
    std::fprintf(f, "%s\n", what);
    std::fclose(f);
  }
}

// The HIP streams of the engines are pooled per process: an engine borrows a set (chain, bulk,
// far) and hands it back drained; streams are created once per (device, CU reservation) and set
// in use at the same time, never destroyed.  Hundreds of engines per process (the test suite, a
// solver that re-analyses) then do not create and destroy hundreds of priority / CU-masked
// streams -- the rare hangs seen on the GPU box sat in runtime calls, not in kernels.
namespace {
struct StreamSet {
  int device, reserve, far_on_bulk;
  hipStream_t chain, bulk, far;
  bool in_use;
  hipStream_t side = nullptr;     // the chain's companion (same priority, no mask): ScheduleOptions::split_next
};
std::mutex g_stream_mu;
std::vector<StreamSet> g_stream_pool;
}  // namespace

static hipError_t borrow_streams(int device, int reserve, int ncu, bool far_on_bulk, hipStream_t* chain,
                                 hipStream_t* bulk, hipStream_t* far, hipStream_t* side) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  for (StreamSet& ss : g_stream_pool)
    if (!ss.in_use && ss.device == device && ss.reserve == reserve && ss.far_on_bulk == (int)far_on_bulk) {
      ss.in_use = true;
      *chain = ss.chain; *bulk = ss.bulk; *far = ss.far; *side = ss.side;
      return hipSuccess;
    }
  int prio_lo = 0, prio_hi = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (e != hipSuccess) return e;
  auto masked_stream = [&](hipStream_t* st) -> hipError_t {
    if (reserve <= 0) return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_lo);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int cu = 0; cu < ncu - reserve; ++cu) mask[cu / 32] |= 1u << (cu % 32);
    return hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data());
  };
  StreamSet ss{device, reserve, (int)far_on_bulk, nullptr, nullptr, nullptr, true};
  // (the chain stream confined to the reserved CUs -- the mirror image of the bulk mask -- was
  // measured: 32.7 ms with 32 CUs, 26.6 with 64, 28.1 with 96 against 23.7: its launches at the
  // lower levels have thousands of workgroups)
  if ((e = hipStreamCreateWithPriority(&ss.chain, hipStreamNonBlocking, prio_hi)) != hipSuccess) return e;
  if ((e = masked_stream(&ss.bulk)) != hipSuccess) return e;
  if (far_on_bulk) ss.far = ss.bulk;
  else if ((e = masked_stream(&ss.far)) != hipSuccess) return e;
  if ((e = hipStreamCreateWithPriority(&ss.side, hipStreamNonBlocking, prio_hi)) != hipSuccess) return e;
  g_stream_pool.push_back(ss);
  *chain = ss.chain; *bulk = ss.bulk; *far = ss.far; *side = ss.side;
  return hipSuccess;
}

// ... and so are the dependency events of the programs (a few hundred per engine, no timing).
// An event taken from the pool may carry the record of an earlier engine: waiting for it is a
// no-op, exactly like waiting for an event that was never recorded.
static std::vector<std::vector<hipEvent_t>> g_event_pool;   // per device

static hipError_t borrow_events(std::vector<hipEvent_t>& out, size_t n, int device) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if ((int)g_event_pool.size() <= device) g_event_pool.resize((size_t)device + 1);
  std::vector<hipEvent_t>& pool = g_event_pool[(size_t)device];
  out.assign(n, nullptr);
  for (size_t i = 0; i < n; ++i) {
    if (!pool.empty()) {
      out[i] = pool.back();
      pool.pop_back();
    } else {
      hipError_t e = hipEventCreateWithFlags(&out[i], hipEventDisableTiming);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

static void return_events(std::vector<hipEvent_t>& ev, int device) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  if ((int)g_event_pool.size() <= device) g_event_pool.resize((size_t)device + 1);
  for (hipEvent_t e : ev)
    if (e) g_event_pool[(size_t)device].push_back(e);
  ev.clear();
}

static void return_streams(hipStream_t chain) {
  std::lock_guard<std::mutex> lk(g_stream_mu);
  for (StreamSet& ss : g_stream_pool)
    if (ss.chain == chain) ss.in_use = false;
}

namespace {
// atexit: the pools go before the HIP runtime does (see the comment at the pools).  Streams that
// are still in use (a leaked, poisoned engine) or not idle are left alone.
void pools_teardown() {
  // SPLLT_TEARDOWN: 0 nothing, 1 (default) everything but the streams, 2 the streams too.
  // hipStreamDestroy of a CU-masked stream is the one call of this library that has been caught
  // not returning (profiles/r03/hang_evidence.txt): it is not made unless asked for -- the
  // profiling scripts ask, because rocprofv3's own teardown crashes on live CU-masked queues,
  // and they run under a timeout.
  static const int mode = [] { const char* e = std::getenv("SPLLT_TEARDOWN"); return e ? std::atoi(e) : 1; }();
  if (mode == 0) return;
  if (g_runtime_wedged.load()) {
    crumb("teardown: skipped, the runtime is wedged");
    return;
  }
  crumb("teardown: streams (lock)");
  // (a helper thread stuck inside borrow_streams / hipExtStreamCreateWithCUMask holds this mutex for good)
  std::unique_lock<std::mutex> lk(g_stream_mu, std::try_to_lock);
  if (!lk.owns_lock()) {
    crumb("teardown: skipped, the stream pool is locked");
    return;
  }
  for (StreamSet& ss : g_stream_pool) {
    if (mode < 2) break;
    if (ss.in_use) continue;
    crumb("teardown: streams (query)");
    if (hipStreamQuery(ss.chain) != hipSuccess || hipStreamQuery(ss.bulk) != hipSuccess ||
        hipStreamQuery(ss.far) != hipSuccess || (ss.side && hipStreamQuery(ss.side) != hipSuccess)) continue;
    crumb("teardown: streams (destroy chain)");
    (void)hipStreamDestroy(ss.chain);
    crumb("teardown: streams (destroy bulk)");
    (void)hipStreamDestroy(ss.bulk);
    crumb("teardown: streams (destroy far)");
    if (ss.far != ss.bulk) (void)hipStreamDestroy(ss.far);
    if (ss.side) (void)hipStreamDestroy(ss.side);      // (plain priority stream: no CU mask)
    ss.chain = ss.bulk = ss.far = ss.side = nullptr;
    ss.in_use = true;        // (never handed out again)
  }
  crumb("teardown: events");
  for (auto& pool : g_event_pool) {
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    pool.clear();
  }
  crumb("teardown: pinned memory");
  {
    std::lock_guard<std::mutex> lk2(g_pinned_mu);
    for (int* q : g_pinned_words) (void)hipHostFree(q);
    g_pinned_words.clear();
  }
  std::lock_guard<std::mutex> lk3(g_pool_mu);
  for (const PinBuf& b : g_pin_cache) (void)hipHostFree(b.p);
  g_pin_cache.clear();
  crumb("teardown: device buffers");
  for (const DevBuf& b : g_dev_cache) (void)hipFree(b.p);
  g_dev_cache.clear();
  g_dev_cached_bytes = 0;
  crumb("teardown: done");
}
}  // namespace

int Engine::upload() {
  const Symbolic& S = *S_;
  crumb("engine: upload begins");
  // The chain and side streams carry the latency-critical kernels: highest priority, all
  // CUs.  The bulk and far streams carry the updates that run BESIDE a chain: lowest
  // priority and masked off the last `reserve_cus` CUs, so that a chain kernel (one
  // workgroup, up to 145 KB of LDS) and the side launches always find a free CU instead of
  // waiting for a bulk workgroup to retire (measured, scripts/cumask_probe.hip: 14 us
  // launch-to-completion beside a masked bulk kernel, 30 us beside an unmasked one; masking
  // the FIRST bits instead gives erratic 13-450 us).  The wide stream (launches that have
  // the chip to themselves) is not masked.
  graph_mode_ = resolve_graph_mode(S, opt_);
  if (const char* e = std::getenv("SPLLT_CHAIN_PRIO")) chain_prio_ = std::atoi(e);
  if (const char* e = std::getenv("SPLLT_BULK_PAD128")) bulk_pad128_ = std::atoi(e);
  if (const char* e = std::getenv("SPLLT_BULK_PAD64")) bulk_pad64_ = std::atoi(e);
  int reserve = opt_.reserve_cus;
  if (const char* e = std::getenv("SPLLT_HIP_RESERVE_CUS")) reserve = std::atoi(e);
  // a factorization that is replayed as a graph has no streams of its own to mask (the kernel nodes
  // of a graph go where the runtime puts them): the small, latency-bound problems -- the ones that
  // used to get CU-masked streams -- no longer create any
  if (graph_mode_ > 0 && opt_.nranks <= 1 && !opt_.poison_lds && !std::getenv("SPLLT_HIP_RESERVE_CUS")) reserve = 0;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_), "device props");
  const int ncu = prop.multiProcessorCount;
  if (reserve < 0 || reserve * 2 > ncu || !opt_.lookahead) reserve = 0;
  // At most four hardware queues carry the five streams of the program: the runtime
  // multiplexes streams onto a handful of hardware queues, and streams that share one
  // serialise (five queues of our own: no overlap at all, 33.0 ms = the serialized 33.2 ms).
  // The wide stream only runs when the chains of a level are done -> chain queue.
  crumb("engine: borrowing streams");
  HIPCHK(borrow_streams(device_, reserve, ncu, std::getenv("SPLLT_FAR_ON_BULK") != nullptr, &streams_[ST_CHAIN],
                        &streams_[ST_BULK], &streams_[ST_FAR], &streams_[ST_SIDE]), "stream creation");
  streams_[ST_WIDE] = streams_[ST_CHAIN];
  if (!streams_[ST_SIDE]) streams_[ST_SIDE] = streams_[ST_CHAIN];
  stream_ = streams_[ST_CHAIN];
  crumb("engine: creating events");
  HIPCHK(borrow_events(dag_events_, (size_t)prog_.nevents, device_), "hipEventCreate");
  crumb("engine: allocating and uploading");
  HIPCHK(hipEventCreate(&ev0_), "hipEventCreate");
  HIPCHK(hipEventCreate(&ev1_), "hipEventCreate");
  HIPCHK(hipEventCreate(&ev_h2d_), "hipEventCreate");
  HIPCHK(hipEventCreateWithFlags(&ev_init_, hipEventDisableTiming), "hipEventCreate");
  // (+ 32 doubles: the update kernel loads whole 16-column chunks, the last one of a ragged K
  // window reaches past the block column's end)
  HIPCHK(dalloc((void**)&d_L_, sizeof(double) * (size_t)(std::max<int64_t>(1, arena_elems_) + 32)), "hipMalloc(L arena)");
  HIPCHK(dalloc((void**)&d_val_, sizeof(double) * (size_t)std::max<int64_t>(1, S.nnzA)), "hipMalloc(val)");
  HIPCHK(dalloc((void**)&d_dinv_, sizeof(double) * (size_t)(std::max<int64_t>(1, prog_.dinv_size) + 32)), "hipMalloc(dinv)");
  // the POTRF kernels store the lower triangle of an inverted panel only; what lies above the
  // diagonal of a slot is zero from here on (nothing else writes there)
  HIPCHK(hipMemset(d_dinv_, 0, sizeof(double) * (size_t)(std::max<int64_t>(1, prog_.dinv_size) + 32)), "hipMemset(dinv)");
  if (prog_.gen_size > 0) {
    // the generated elements of the subtree tasks: zero here, and zero again after every factorization
    // (the root of a subtree clears what it adds to its ancestors)
    HIPCHK(dalloc((void**)&d_gen_, sizeof(double) * (size_t)prog_.gen_size), "hipMalloc(generated elements)");
    HIPCHK(hipMemset(d_gen_, 0, sizeof(double) * (size_t)prog_.gen_size), "hipMemset(generated elements)");
  }
  if (opt_.nranks > 1) {
    // this rank scatters A only into its own subtrees; the top tree's values
    // are contributed by rank 0 alone so that the cross-rank sum holds them once
    std::vector<int64_t> md, ms;
    map_keep_.assign(S.map_dst.size(), 0);
    for (int b = 0; b < S.nbcol(); ++b) {
      const int own = owner_[S.bcols[b].node];
      const bool keep = (own == opt_.rank) || (own < 0 && opt_.rank == 0);
      if (!keep) continue;
      const int64_t delta = loc_off_[(size_t)b] - S.bcols[b].off;   // (kept block columns are held here)
      for (int64_t i = S.lmap_ptr[b]; i < S.lmap_ptr[b + 1]; ++i) {
        map_keep_[i] = 1;
        md.push_back(S.map_dst[i] + delta);
        ms.push_back(S.map_src[i]);
      }
    }
    nmap_ = (int64_t)md.size();
    HIPCHK(dev_upload(&d_map_dst_, md), "upload map_dst");
    HIPCHK(dev_upload(&d_map_src_, ms), "upload map_src");
  } else {
    nmap_ = S.nnzA;
    HIPCHK(dev_upload(&d_map_dst_, S.map_dst), "upload map_dst");
    HIPCHK(dev_upload(&d_map_src_, S.map_src), "upload map_src");
    // the same map bucketed by 32 KB chunks of the arena, for the one-pass initialisation
    // (k_init_arena): a counting sort, two passes over the map
    static const bool one_pass = [] { const char* e = std::getenv("SPLLT_INIT_ONE_PASS"); return !(e && std::atoi(e) == 0); }();
    if (one_pass && S.arena > 0 && S.nnzA <= INT_MAX) {
      const int64_t nch = (S.arena + kInitChunk - 1) / kInitChunk;
      std::vector<int64_t> cptr((size_t)nch + 1, 0);
      for (int64_t d : S.map_dst) cptr[(size_t)(d / kInitChunk) + 1]++;
      for (int64_t c = 0; c < nch; ++c) cptr[(size_t)c + 1] += cptr[(size_t)c];
      std::vector<int64_t> fill(cptr.begin(), cptr.end() - 1);
      std::vector<unsigned short> loc(S.map_dst.size());
      std::vector<int> src(S.map_dst.size());
      for (size_t i = 0; i < S.map_dst.size(); ++i) {
        const int64_t d = S.map_dst[i];
        const int64_t at = fill[(size_t)(d / kInitChunk)]++;
        loc[(size_t)at] = (unsigned short)(d % kInitChunk);
        src[(size_t)at] = (int)S.map_src[i];
      }
      HIPCHK(dev_upload(&d_init_cptr_, cptr), "upload init map");
      HIPCHK(dev_upload(&d_init_loc_, loc), "upload init map");
      HIPCHK(dev_upload(&d_init_src_, src), "upload init map");
    }
  }
  std::vector<int64_t> off(S.nbcol());
  std::vector<int> w(S.nbcol());
  for (int b = 0; b < S.nbcol(); ++b) {
    off[b] = loc_off_.empty() ? S.bcols[b].off : loc_off_[(size_t)b];   // (-1: never dereferenced here)
    w[b] = S.bcols[b].width;
  }
  TableStager tab;
  tab.add(&d_bc_off_, off);
  tab.add(&d_bc_w_, w);
  std::vector<UpdUnit> units;    // (outlives the staging copy below)
  if (prog_.scratch_size > 0) {
    HIPCHK(dalloc((void**)&d_scratch_, sizeof(double) * (size_t)prog_.scratch_size), "hipMalloc(scratch)");
    // MODE_BUFFER units address the scratch relative to the arena pointer like every other unit
    units = prog_.units;
    const int64_t shift = d_scratch_ - d_L_;
    for (UpdUnit& u : units)
      if (u.mode == MODE_BUFFER) u.d_off += shift;
    tab.add(&d_units_, units);
  } else {
    tab.add(&d_units_, prog_.units);
  }
  tab.add(&d_gtiles_, prog_.gather_tiles);
  tab.add(&d_gitems_, prog_.gather_items);
  tab.add(&d_tiles_, prog_.tiles);
  tab.add(&d_chain_, prog_.chain_units);
  tab.add(&d_panel_, prog_.panel_units);
  tab.add(&d_sub_tasks_, prog_.sub_tasks);
  tab.add(&d_sub_nodes_, prog_.sub_nodes);
  const std::vector<int> zeros(2 * std::max<size_t>(1, prog_.panel_units.size()), 0);
  tab.add(&d_panel_cnt_, zeros);
  tab.add(&d_relpos_, prog_.relpos);
  tab.add(&d_rlist_, S.rlist);
  const int big_flag = INT_MAX;
  tab.add(&d_flag_, &big_flag, 1);
  HIPCHK(tab.commit(&d_tables_, [this](void** q, size_t b) { return dalloc(q, b); }), "upload tables");
  HIPCHK(borrow_pinned_word(&h_flag_), "hipHostMalloc(flag)");
  crumb("engine: ready");
  return 0;
}

// Arena offset of the global layout -> this rank's packed arena.  Offsets in the tables are
// always the base of a block column.
int64_t Engine::to_local(int64_t g) const {
  const auto& bc = S_->bcols;
  size_t lo = 0, hi = bc.size();
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (bc[mid].off <= g) lo = mid; else hi = mid;
  }
  if (bc[lo].off != g || loc_off_[lo] < 0) {
    // (const: called from a const context too; the flag is picked up by localize_program)
    std::fprintf(stderr, "spllt-hip: internal error: arena offset %lld is not a block column held by rank %d\n",
                 (long long)g, opt_.rank);
    localize_failed_ = true;
    return 0;
  }
  return loc_off_[lo];
}

void Engine::localize_program() {
  const Symbolic& S = *S_;
  loc_off_.assign((size_t)S.nbcol(), -1);
  int64_t o = 0;
  for (int b = 0; b < S.nbcol(); ++b) {
    const int own = owner_[S.bcols[b].node];
    if (own != opt_.rank && own >= 0) continue;
    loc_off_[(size_t)b] = o;
    o += (int64_t)S.bcols[b].nrow * S.bcols[b].width;
  }
  arena_elems_ = o;
  zero_ranges_.assign(1, {0, o});      // everything held here is cleared per factorization
  for (UpdUnit& u : prog_.units) {
    u.a_off = to_local(u.a_off);
    if (u.mode != MODE_BUFFER) u.d_off = to_local(u.d_off);   // (BUFFER: an offset into the scratch)
  }
  for (ChainUnit& u : prog_.chain_units) u.off = to_local(u.off);
  for (PanelUnit& u : prog_.panel_units) u.off = to_local(u.off);
  for (GatherTile& t : prog_.gather_tiles) t.d_off = to_local(t.d_off);
  for (ExchangeItem& it : prog_.xitems)
    if (it.space == 0) it.off = to_local(it.off);
  if (localize_failed_) {
    // a table entry points at a block column this rank does not hold: running the program would
    // silently read and write this rank's first block column instead
    status_ = -30;
    err_ = "internal error: the rank's program refers to a block column it does not hold";
  }
}

Engine::~Engine() {
  if (graph_exec_ && !poisoned_) hipGraphExecDestroy(graph_exec_);
  if (graph_ && !poisoned_) hipGraphDestroy(graph_);
  if (poisoned_) {
    // A wait of this engine ran into its deadline (or its submission never returned): the device
    // may still execute, or be stuck in, what it enqueued.  Nothing is synchronised (that would
    // block for good), nothing goes back into the pools (the next engine would run behind, or
    // beside, the stale program), nothing the queued work may touch is freed: it is leaked.
    crumb("engine: poisoned, resources leaked");
    return;
  }
  crumb("engine: destructor, draining streams");
  for (hipStream_t st : streams_)
    if (st && sync_stream(st, "engine teardown") != 0) {
      crumb("engine: teardown ran into the deadline, resources leaked");
      return;                               // (poisoned by sync_stream)
    }
  crumb("engine: destructor, releasing");
  return_events(dag_events_, device_);       // (the streams are drained: nothing refers to them any more)
  for (const auto& b : owned_) dev_release(b.first, b.second, device_);   // kept for the next engine, not hipFree'd:
  owned_.clear();                                                          // hipFree synchronises the whole device
  if (h2d_buf_[0]) pin_release(h2d_buf_[0], h2d_chunk_);
  if (h2d_buf_[1]) pin_release(h2d_buf_[1], h2d_chunk_);
  for (hipEvent_t e : h2d_ev_)
    if (e) hipEventDestroy(e);
  return_pinned_word(h_flag_);
  if (ev0_) hipEventDestroy(ev0_);
  if (ev1_) hipEventDestroy(ev1_);
  if (ev_h2d_) hipEventDestroy(ev_h2d_);
  if (ev_init_) hipEventDestroy(ev_init_);
  if (streams_[ST_CHAIN]) return_streams(streams_[ST_CHAIN]);   // drained above; back into the pool
  crumb("engine: destroyed");
}

// the kernel of one launch of the program, onto a stream or into a graph (LaunchSink)
void Engine::emit_kernel(const Launch& l, const LaunchSink& sink, bool multi) {
  if (l.kind == L_CHAIN) {
    launch_chain_panel(sink, d_chain_ + l.first, l.count, d_L_, d_dinv_, d_flag_, prog_.chain_units[(size_t)l.first]);
  } else if (l.kind == L_CHAIN4) {
    launch_chain_block(sink, d_chain_ + l.first, l.count, d_L_, d_dinv_, d_flag_, prog_.pw,
                       prog_.chain_units[(size_t)l.first]);
  } else if (l.kind == L_TRSM4) {
    launch_trsm_rows(sink, d_tiles_ + l.first, l.count, d_units_, d_L_, d_dinv_, prog_.pw,
                     (multi && l.stream == ST_CHAIN) ? chain_prio_ : 0);
  } else if (l.kind == L_SUBTREE) {
    launch_subtree(sink, d_sub_tasks_ + l.first, l.count, d_sub_nodes_, d_units_, d_relpos_, d_rlist_, d_L_, d_dinv_,
                   d_gen_, d_flag_);
  } else if (l.kind == L_PANEL) {
    launch_panel(sink, d_tiles_ + l.first, l.count, d_panel_, d_L_, d_dinv_, d_panel_cnt_, d_flag_);
  } else if (l.kind == L_GATHER) {
    launch_gather(sink, d_gtiles_ + l.first, l.count, d_gitems_, d_L_, d_scratch_, d_relpos_, d_rlist_);
  } else {
    // multi-stream program: chain / side launches run at raised wave priority
    const int prio = (multi && (l.stream == ST_CHAIN || l.stream == ST_SIDE)) ? chain_prio_ : 0;
    int pad = 0;
    if (multi && l.overlap) pad = l.tile == 128 ? bulk_pad128_ : bulk_pad64_;
    launch_update(sink, l.tile, d_tiles_ + l.first, l.count, d_units_, d_bc_off_, d_bc_w_, d_L_,
                  d_relpos_, d_rlist_, d_dinv_, prio, pad, true, l.lat != 0);
  }
}

int Engine::enqueue_launch(const Launch& l, bool serial) {
  hipStream_t st = serial ? stream_ : streams_[l.stream];
  if (!serial)
    for (int w : l.wait)
      if (w >= 0) HIPCHK(hipStreamWaitEvent(st, dag_events_[w], 0), "stream wait");
  if (l.count > 0 && l.kind != L_EXCHANGE) {
    if (opt_.poison_lds) launch_poison_lds(st);
    emit_kernel(l, LaunchSink(st), !serial && opt_.lookahead);
  }
  if (!serial && l.record >= 0) HIPCHK(hipEventRecord(dag_events_[l.record], st), "event record");
  return 0;
}

// The whole factorization of one pattern as ONE HIP graph (analyse once,
// factorize many -- reference spllt_kernels_mod.F90:2301-2364 re-initialises the same structure
// for new values).  Built explicitly from the program (no stream capture: capturing the
// multi-stream program crashed inside the runtime on ROCm 7.2): clear the arena, the "last
// reader" counters and the flag, scatter the values, then one kernel node per launch, the flag's
// way back to the host last.  Every address is fixed for the life of the engine; only the
// content of d_val_ changes between replays.
//   mode 1: a chain in program order (= the single-stream program: every node behind the one in
//           front; 1.5 us per boundary instead of the eager path's 2.7 us, or 6.4 us where an
//           event record sits between two launches -- scripts/gap_probe.hip)
//   mode 2: the DAG of the multi-stream program: a node follows its predecessor on the same
//           program stream and the launches that record the events it waits for
int Engine::build_graph(int mode) {
  if (graph_exec_) return 0;
  if (!prog_.exchanges.empty() || opt_.poison_lds) return -98;   // single-GPU programs only
  const Symbolic& S = *S_;
  HIPCHK(hipGraphCreate(&graph_, 0), "hipGraphCreate");
  std::vector<hipGraphNode_t> head;      // the prefix every launch follows
  auto memset_node = [&](void* dst, int value, size_t words, const std::vector<hipGraphNode_t>& deps,
                         hipGraphNode_t* out) -> hipError_t {
    hipMemsetParams mp{};
    mp.dst = dst;
    mp.elementSize = 4;
    mp.value = (unsigned)value;
    mp.width = words;
    mp.height = 1;
    mp.pitch = words * 4;
    return hipGraphAddMemsetNode(out, graph_, deps.data(), deps.size(), &mp);
  };
  hipGraphNode_t n_cnt = nullptr, n_flag = nullptr;
  std::vector<hipGraphNode_t> pre;
  const bool one_pass_init = d_init_cptr_ != nullptr;     // k_init_arena clears as it copies
  if (!one_pass_init) {
    // (in pieces of 4 GiB: 32-bit element counts somewhere below would not be a surprise)
    const size_t words = (size_t)S.arena * 2, piece = (size_t)1 << 30;
    for (size_t o = 0; o < words; o += piece) {
      hipGraphNode_t n = nullptr;
      HIPCHK(memset_node((int*)d_L_ + o, 0, std::min(piece, words - o), {}, &n), "graph memset arena");
      pre.push_back(n);
    }
  }
  if (!prog_.panel_units.empty()) {
    HIPCHK(memset_node(d_panel_cnt_, 0, 2 * prog_.panel_units.size(), {}, &n_cnt), "graph memset counters");
    pre.push_back(n_cnt);
  }
  HIPCHK(memset_node(d_flag_, INT_MAX, 1, {}, &n_flag), "graph memset flag");
  pre.push_back(n_flag);
  hipGraphNode_t n_scatter = nullptr;
  {
    LaunchSink sink;
    sink.graph = graph_;
    sink.deps = pre.data();
    sink.ndeps = pre.size();
    if (one_pass_init)
      launch_init_arena(sink, d_L_, S.arena, d_val_, d_init_cptr_, d_init_loc_, d_init_src_);
    else
      launch_scatter_val(sink, d_L_, d_val_, d_map_dst_, d_map_src_, nmap_);
    HIPCHK(sink.err, "graph scatter node");
    n_scatter = sink.node;
    if (!n_scatter) {     // (no entries: an empty node keeps the prefix in one piece)
      HIPCHK(hipGraphAddEmptyNode(&n_scatter, graph_, pre.data(), pre.size()), "graph empty node");
    }
  }
  const size_t nl = prog_.launches.size();
  std::vector<hipGraphNode_t> recorder((size_t)std::max(1, prog_.nevents), nullptr);   // event id -> node
  hipGraphNode_t last_on[ST_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipGraphNode_t last = n_scatter;
  for (size_t i = 0; i < nl; ++i) {
    const Launch& l = prog_.launches[i];
    std::vector<hipGraphNode_t> deps;
    auto add_dep = [&](hipGraphNode_t n) {
      if (!n) return;
      for (hipGraphNode_t d : deps)
        if (d == n) return;
      deps.push_back(n);
    };
    if (mode == 1) {
      add_dep(last);
    } else {
      // (the wide and side streams are the chain stream, as in the eager engine)
      const int sid = (l.stream == ST_WIDE || l.stream == ST_SIDE) ? ST_CHAIN : l.stream;
      add_dep(last_on[sid] ? last_on[sid] : n_scatter);
      for (int w : l.wait)
        if (w >= 0) add_dep(recorder[(size_t)w]);
    }
    hipGraphNode_t node = nullptr;
    if (l.count > 0) {
      LaunchSink sink;
      sink.graph = graph_;
      sink.deps = deps.data();
      sink.ndeps = deps.size();
      emit_kernel(l, sink, mode == 2 && opt_.lookahead);
      HIPCHK(sink.err, "graph kernel node");
      node = sink.node;
    } else {
      HIPCHK(hipGraphAddEmptyNode(&node, graph_, deps.data(), deps.size()), "graph marker node");
    }
    last = node;
    const int sid = (l.stream == ST_WIDE || l.stream == ST_SIDE) ? ST_CHAIN : l.stream;
    last_on[sid] = node;
    if (l.record >= 0) recorder[(size_t)l.record] = node;
  }
  {
    std::vector<hipGraphNode_t> deps;
    if (mode == 1) {
      deps.push_back(last);
    } else {
      auto add = [&](hipGraphNode_t n) {
        if (n && std::find(deps.begin(), deps.end(), n) == deps.end()) deps.push_back(n);
      };
      for (hipGraphNode_t n : last_on) add(n);
      if (prog_.final_event >= 0) add(recorder[(size_t)prog_.final_event]);
      if (deps.empty()) deps.push_back(n_scatter);
    }
    hipGraphNode_t n_out = nullptr;
    HIPCHK(hipGraphAddMemcpyNode1D(&n_out, graph_, deps.data(), deps.size(), h_flag_, d_flag_, sizeof(int),
                                   hipMemcpyDeviceToHost), "graph flag read");
  }
  HIPCHK(hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0), "hipGraphInstantiate");
  return 0;
}

int Engine::enqueue_range(size_t first, size_t last) {
  for (size_t i = first; i < last; ++i) {
    int rc = enqueue_launch(prog_.launches[i], false);
    if (rc) return rc;
  }
  return 0;
}

int Engine::finish_enqueue() {
  if (prog_.final_event >= 0)
    HIPCHK(hipStreamWaitEvent(stream_, dag_events_[prog_.final_event], 0), "final wait");
  HIPCHK(hipGetLastError(), "kernel launch");
  HIPCHK(hipMemcpyAsync(h_flag_, d_flag_, sizeof(int), hipMemcpyDeviceToHost, stream_), "flag read");
  return 0;
}

int Engine::enqueue_program() {
  const Symbolic& S = *S_;
  if (replays_graph()) {
    int rc = build_graph(graph_mode_);
    if (rc) return rc;
    stats_.launches = (int)prog_.launches.size() + 1;
    HIPCHK(hipGraphLaunch(graph_exec_, stream_), "hipGraphLaunch");
    awaiting_exchange_ = false;
    return 0;
  }
  const bool one_pass_init = opt_.nranks <= 1 && d_init_cptr_ != nullptr;
  if (opt_.nranks > 1) {
    for (const auto& r : zero_ranges_)
      HIPCHK(hipMemsetAsync(d_L_ + r.first, 0, sizeof(double) * (size_t)r.second, stream_), "memset arena");
  } else if (!one_pass_init) {
    HIPCHK(hipMemsetAsync(d_L_, 0, sizeof(double) * (size_t)S.arena, stream_), "memset arena");
  }
  // the "last reader" counters of the fused panel launches return to zero by themselves; a
  // factorization that was cut short (a failed launch, the watchdog) may have left some behind
  if (!prog_.panel_units.empty())
    HIPCHK(hipMemsetAsync(d_panel_cnt_, 0, sizeof(int) * 2 * prog_.panel_units.size(), stream_), "memset counters");
  const int big = INT_MAX;
  *h_flag_ = big;
  HIPCHK(hipMemcpyAsync(d_flag_, h_flag_, sizeof(int), hipMemcpyHostToDevice, stream_), "flag init");
  if (one_pass_init)
    launch_init_arena(stream_, d_L_, S.arena, val_src_ ? val_src_ : d_val_, d_init_cptr_, d_init_loc_, d_init_src_);
  else
    launch_scatter_val(stream_, d_L_, val_src_ ? val_src_ : d_val_, d_map_dst_, d_map_src_, nmap_);
  if (!prog_.exchanges.empty() || !prog_.sub_tasks.empty()) {
    // (the subtree tasks are the first launch of the side stream and wait for nothing else.)
    // A partitioned program may have an exchange as its FIRST launch on a stream other than this one
    // (a rank that owns no subtree: its phase 1 is empty, and the per-level reduce-scatters of a
    // distributed top tree run on the side stream): nothing would order its pack behind the clearing
    // of the arena and the scatter of the values above.  Every other stream starts behind them.
    HIPCHK(hipEventRecord(ev_init_, stream_), "event");
    for (int i = 0; i < ST_COUNT; ++i)
      if (streams_[i] && streams_[i] != stream_) HIPCHK(hipStreamWaitEvent(streams_[i], ev_init_, 0), "init wait");
  }
  stats_.launches = (int)prog_.launches.size() + 1;
  if (!prog_.exchanges.empty() && !xbuf_) return fail(-10, "exchange buffer not set", hipSuccess);
  return run_from(0);
}

// Launches [first, ...) until the next exchange point (packed, awaiting_exchange_) or the end.
int Engine::run_from(size_t first) {
  for (size_t i = first; i < prog_.launches.size(); ++i) {
    const Launch& l = prog_.launches[i];
    if (l.kind == L_EXCHANGE) {
      int rc = pre_exchange(l);
      if (rc) return rc;
      cur_x_ = i;
      awaiting_exchange_ = true;
      return 0;
    }
    int rc = enqueue_launch(l, false);
    if (rc) return rc;
  }
  awaiting_exchange_ = false;
  return finish_enqueue();
}

// The part of an exchange in front of the collective: its dependencies, then what this rank
// contributes goes into the exchange buffer (all on the chain stream, where the caller then
// enqueues the collective).
// the stream an exchange runs on: pack, collective and unpack (the chain stream, or -- the per-level
// reduce-scatters of a distributed top tree in the multi-stream program -- the side stream, so that
// the chunks of the upper levels travel while the lowest top level is already being factorized)
hipStream_t Engine::exchange_stream(const Launch& X) const {
  return (X.stream >= 0 && X.stream < ST_COUNT && streams_[X.stream]) ? streams_[X.stream] : stream_;
}

int Engine::pre_exchange(const Launch& X) {
  const Exchange& E = prog_.exchanges[(size_t)X.first];
  hipStream_t xs = exchange_stream(X);
  for (int w : X.wait)
    if (w >= 0) HIPCHK(hipStreamWaitEvent(xs, dag_events_[w], 0), "exchange wait");
  for (int i = E.first_item; i < E.first_item + E.nitems; ++i) {
    const ExchangeItem& it = prog_.xitems[(size_t)i];
    if (E.kind == X_BCAST && it.root != opt_.rank) continue;   // the owner sends
    HIPCHK(hipMemcpyAsync(xbuf_ + it.xoff, (it.space ? d_dinv_ : d_L_) + it.off, sizeof(double) * (size_t)it.count,
                          hipMemcpyDeviceToDevice, xs), "pack exchange");
  }
  if (E.kind == X_REDUCE_ALL || E.kind == X_FLAG) launch_flag_pack(xs, d_flag_, xbuf_ + (E.elems - 1));
  HIPCHK(hipGetLastError(), "kernel launch");
  return 0;
}

// ... and behind it: what this rank needs comes out of the buffer, the exchange's event fires.
int Engine::post_exchange(const Launch& X) {
  const Exchange& E = prog_.exchanges[(size_t)X.first];
  hipStream_t xs = exchange_stream(X);
  for (int i = E.first_item; i < E.first_item + E.nitems; ++i) {
    const ExchangeItem& it = prog_.xitems[(size_t)i];
    if (E.kind == X_BCAST && it.root == opt_.rank) continue;          // already here
    if (E.kind == X_REDUCE_OWNER && it.root != opt_.rank) continue;   // somebody else's sum
    HIPCHK(hipMemcpyAsync((it.space ? d_dinv_ : d_L_) + it.off, xbuf_ + it.xoff, sizeof(double) * (size_t)it.count,
                          hipMemcpyDeviceToDevice, xs), "unpack exchange");
  }
  if (E.kind == X_REDUCE_ALL || E.kind == X_FLAG) launch_flag_unpack(xs, xbuf_ + (E.elems - 1), d_flag_);
  if (X.record >= 0) HIPCHK(hipEventRecord(dag_events_[X.record], xs), "exchange record");
  return 0;
}

// ---------------------------------------------------------------------------
// RCCL, resolved at run time from the librccl the process already has (the caller created the
// communicator with it; a Python process has torch's copy): no link-time dependency, one copy.
// ---------------------------------------------------------------------------
namespace {
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_reducescatter_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_broadcast_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_group_t)();
typedef int (*nccl_query_t)(void*, int*);
typedef const char* (*nccl_errstr_t)(int);
struct Rccl {
  void* handle = nullptr;
  nccl_allreduce_t all_reduce = nullptr;
  nccl_reducescatter_t reduce_scatter = nullptr;
  nccl_broadcast_t broadcast = nullptr;
  nccl_group_t group_start = nullptr, group_end = nullptr;
  nccl_query_t comm_count = nullptr, comm_user_rank = nullptr;
  nccl_errstr_t err_string = nullptr;
  bool ok() const { return all_reduce && reduce_scatter && broadcast && group_start && group_end && comm_count && comm_user_rank; }
};
constexpr int kNcclDouble = 8, kNcclSum = 0;      // ncclFloat64, ncclSum (rccl.h)
Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    const char* names[] = {"librccl.so.1", "librccl.so", "libnccl.so.2"};
    for (const char* nm : names)
      if ((q.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;     // the copy the process already uses
    if (!q.handle)
      for (const char* nm : names)
        if ((q.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!q.handle) return q;
    q.all_reduce = (nccl_allreduce_t)dlsym(q.handle, "ncclAllReduce");
    q.reduce_scatter = (nccl_reducescatter_t)dlsym(q.handle, "ncclReduceScatter");
    q.broadcast = (nccl_broadcast_t)dlsym(q.handle, "ncclBroadcast");
    q.group_start = (nccl_group_t)dlsym(q.handle, "ncclGroupStart");
    q.group_end = (nccl_group_t)dlsym(q.handle, "ncclGroupEnd");
    q.comm_count = (nccl_query_t)dlsym(q.handle, "ncclCommCount");
    q.comm_user_rank = (nccl_query_t)dlsym(q.handle, "ncclCommUserRank");
    q.err_string = (nccl_errstr_t)dlsym(q.handle, "ncclGetErrorString");
    return q;
  }();
  return r;
}
}  // namespace

#define NCCLCHK(call, what)                                                                      \
  do {                                                                                           \
    int r__ = (call);                                                                            \
    if (r__ != 0) {                                                                              \
      status_ = kErrHip;                                                                         \
      err_ = std::string(what) + ": RCCL error " + std::to_string(r__) +                         \
             (rccl().err_string ? std::string(" (") + rccl().err_string(r__) + ")" : std::string()); \
      std::fprintf(stderr, "spllt-hip: %s\n", err_.c_str());                                     \
      return kErrHip;                                                                            \
    }                                                                                            \
  } while (0)

int Engine::set_communicator(void* nccl_comm) {
  if (status_) return status_;
  if (!nccl_comm) { comm_ = nullptr; return 0; }
  Rccl& R = rccl();
  if (!R.ok()) return fail(-10, "spllt_hip_set_communicator: librccl is not loadable", hipSuccess);
  int cnt = 0, rk = 0;
  NCCLCHK(R.comm_count(nccl_comm, &cnt), "ncclCommCount");
  NCCLCHK(R.comm_user_rank(nccl_comm, &rk), "ncclCommUserRank");
  // SPLLT_HIP_COMM_REHEARSAL=1: a communicator smaller than the partition is accepted (one-GPU
  // rehearsal of the call sequence: the sums then miss the other ranks' parts, broadcast roots
  // are taken modulo its size)
  // -- accepted only for a ONE-rank communicator (the one-GPU test of the call sequence), announced
  // on stderr, and remembered: spllt_hip_last_error says so for as long as the handle lives.
  static const bool rehearsal_env = std::getenv("SPLLT_HIP_COMM_REHEARSAL") != nullptr;
  const bool mismatch = cnt != opt_.nranks || rk != opt_.rank;
  const bool rehearsal = rehearsal_env && cnt == 1;
  if (mismatch && rehearsal) {
    comm_rehearsal_ = true;
    std::fprintf(stderr, "spllt-hip: WARNING: SPLLT_HIP_COMM_REHEARSAL: a one-rank communicator stands in for rank %d of %d "
                         "-- the collectives run, the factor of this handle is NOT the factor of the matrix\n",
                 opt_.rank, opt_.nranks);
  }
  if (mismatch && !rehearsal) {
    err_ = "spllt_hip_set_communicator: the communicator is rank " + std::to_string(rk) + " of " + std::to_string(cnt) +
           ", the partition (spllt_hip_set_partition) is rank " + std::to_string(opt_.rank) + " of " +
           std::to_string(opt_.nranks);
    std::fprintf(stderr, "spllt-hip: %s\n", err_.c_str());
    return -10;
  }
  comm_ = nccl_comm;
  comm_rank_ = rk;
  comm_size_ = cnt;
  if (!xbuf_ && prog_.xbuf_elems > 0) {      // the exchange buffer is the library's own in this mode
    HIPCHK(dalloc((void**)&xbuf_, sizeof(double) * (size_t)prog_.xbuf_elems), "hipMalloc(exchange buffer)");
    HIPCHK(hipMemset(xbuf_, 0, sizeof(double) * (size_t)prog_.xbuf_elems), "hipMemset(exchange buffer)");
  }
  if (opt_.nranks > 1 && !d_owned_) {
    // a rank's share of a distributed vector: its own subtrees; rank 0 also the top tree
    std::vector<double> keep((size_t)S_->n, 0.0);
    for (int s = 0; s < S_->nnodes; ++s) {
      const int own = owner_[(size_t)s];
      const bool mine = own == opt_.rank || (own < 0 && opt_.rank == 0);
      if (mine)
        for (int p = S_->sptr[s]; p < S_->sptr[s + 1]; ++p) keep[(size_t)p] = 1.0;
    }
    HIPCHK(dev_upload(&d_owned_, keep), "upload owner mask");
  }
  return 0;
}

// the collective of one exchange on the exchange buffer, in place, enqueued on the engine's stream
// (between pre_exchange's pack and post_exchange's unpack)
int Engine::collective(const Exchange& E, hipStream_t xs) {
  Rccl& R = rccl();
  if (E.kind == X_REDUCE_ALL || E.kind == X_FLAG) {
    NCCLCHK(R.all_reduce(xbuf_, xbuf_, (size_t)E.elems, kNcclDouble, kNcclSum, comm_, xs), "ncclAllReduce");
  } else if (E.kind == X_REDUCE_OWNER) {
    // in place: rank r receives into its own chunk of the region (the region of this level's
    // exchange ends at E.elems: schedule.cpp)
    double* region = xbuf_ + (E.elems - (int64_t)opt_.nranks * E.chunk);
    NCCLCHK(R.reduce_scatter(region, region + (int64_t)comm_rank_ * E.chunk, (size_t)E.chunk, kNcclDouble, kNcclSum,
                             comm_, xs), "ncclReduceScatter");
  } else if (E.kind == X_BCAST) {
    // one broadcast per root (the items of a root are contiguous in the buffer), as one group
    // (a failed broadcast does not leave the group open: ncclGroupEnd is always reached -- the
    // communicator would stay in group mode and the other ranks in the collective -- and the first
    // error is reported afterwards)
    NCCLCHK(R.group_start(), "ncclGroupStart");
    int first_err = 0;
    for (int i = E.first_item; i < E.first_item + E.nitems;) {
      const ExchangeItem& a = prog_.xitems[(size_t)i];
      int64_t cnt = 0;
      int j = i;
      for (; j < E.first_item + E.nitems && prog_.xitems[(size_t)j].root == a.root &&
             prog_.xitems[(size_t)j].xoff == a.xoff + cnt; ++j)
        cnt += prog_.xitems[(size_t)j].count;
      const int r = R.broadcast(xbuf_ + a.xoff, xbuf_ + a.xoff, (size_t)cnt, kNcclDouble, a.root % comm_size_, comm_,
                                xs);
      if (r != 0 && first_err == 0) first_err = r;
      i = j;
    }
    const int rend = R.group_end();
    NCCLCHK(first_err, "ncclBroadcast");
    NCCLCHK(rend, "ncclGroupEnd");
  }
  return 0;
}

int Engine::run_exchanges() {
  if (status_) return status_;
  while (awaiting_exchange_) {
    if (!comm_) return 0;                     // the caller drives the exchanges (spllt_hip_continue)
    const Exchange& E = prog_.exchanges[(size_t)prog_.launches[cur_x_].first];
    int rc = collective(E, exchange_stream(prog_.launches[cur_x_]));
    if (rc) return rc;
    rc = continue_after_exchange();
    if (rc) return rc;
  }
  return 0;
}

int Engine::sync_phase() {
  if (status_) return status_;
  for (hipStream_t st : streams_)
    if (st) {
      int rc = sync_stream(st, "stream sync");
      if (rc) return rc;
    }
  return 0;
}

int Engine::continue_after_exchange() {
  if (status_) return status_;
  if (!awaiting_exchange_) return -10;
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  int rc = post_exchange(prog_.launches[cur_x_]);
  if (rc) return rc;
  rc = run_from(cur_x_ + 1);
  if (rc) return rc;
  if (!awaiting_exchange_) HIPCHK(hipEventRecord(ev1_, stream_), "event");
  return 0;
}

int Engine::factor_async_dev(const double* val_dev, int64_t nnz) {
  if (status_) return status_;
  if (nnz != S_->nnzA) return -10;
  double t0 = now_ms();
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  HIPCHK(hipEventRecord(ev0_, stream_), "event");
  // Eager launches read the caller's device array directly (it must stay valid and unchanged until
  // spllt_hip_wait: include/spllt_hip.h) -- the copy into the engine's own buffer was 88 MB each way
  // on the bench workload, 60 us in front of every factorization.  A graph replay has the engine's
  // buffer baked into its nodes: there the values are copied.
  val_src_ = nullptr;
  if (val_dev != d_val_) {
    if (replays_graph())
      HIPCHK(hipMemcpyAsync(d_val_, val_dev, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, stream_), "val D2D");
    else
      val_src_ = val_dev;
  }
  HIPCHK(hipEventRecord(ev_h2d_, stream_), "event");
  int rc = enqueue_program();
  if (rc) return rc;
  if (comm_ && (rc = run_exchanges())) return rc;
  HIPCHK(hipEventRecord(ev1_, stream_), "event");
  pending_ = true;
  stats_.submit_ms = now_ms() - t0;
  return 0;
}

int Engine::factor_async(const double* val_host, int64_t nnz) {
  if (status_) return status_;
  if (nnz != S_->nnzA) return -10;
  double t0 = now_ms();
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  HIPCHK(hipEventRecord(ev0_, stream_), "event");
  val_src_ = nullptr;            // (the values go through the engine's own buffer)
  crumb("factor: H2D of val");
  // test hook (tests/test_gpu_parity.py::test_submission_deadline): a runtime call that sits
  if (const char* e = std::getenv("SPLLT_HIP_TEST_STALL_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(e)));
  static const bool direct_h2d = [] { const char* e = std::getenv("SPLLT_HIP_H2D"); return e && std::string(e) == "direct"; }();
  if (direct_h2d) {
    HIPCHK(hipMemcpyAsync(d_val_, val_host, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, stream_), "val H2D");
  } else {
    int rc = stage_val(val_host, nnz);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(ev_h2d_, stream_), "event");
  crumb("factor: enqueueing the program");
  int rc = enqueue_program();
  if (rc) return rc;
  if (comm_ && (rc = run_exchanges())) return rc;
  crumb("factor: enqueued");
  HIPCHK(hipEventRecord(ev1_, stream_), "event");
  pending_ = true;
  stats_.submit_ms = now_ms() - t0;
  return 0;
}

int Engine::wait() {
  if (status_) return status_;
  if (!pending_) return 0;
  crumb("wait: polling the chain stream");
  if (awaiting_exchange_) return sync_phase();  // not finished: only drain phase 1
  {
    int rc = sync_stream(stream_, "stream sync");
    if (rc) return rc;
  }
  pending_ = false;
  crumb("wait: done");
  float ms = 0;
  if (hipEventElapsedTime(&ms, ev0_, ev1_) == hipSuccess) stats_.device_ms = ms;
  if (hipEventElapsedTime(&ms, ev0_, ev_h2d_) == hipSuccess) stats_.h2d_ms = ms;
  npd_col_ = -1;
  if (*h_flag_ != INT_MAX) {
    npd_col_ = *h_flag_ == INT_MAX - 1 ? -1 : *h_flag_ - 1;   // -1: reported by another rank
    return kErrNotPosDef;
  }
  return 0;
}

// device -> pageable host memory through the two pinned staging buffers of stage_val (the mirror
// image: the DMA of chunk k + 1 flies while the host copies chunk k out; every wait has the
// deadline).  A plain hipMemcpy into pageable memory moved the 1.5 GB factor of the bench workload
// at 3.2 GB/s (the runtime stages it itself, one small buffer at a time).
int Engine::staged_d2h(void* out_host, const void* src_dev, size_t bytes) {
  if (bytes == 0) return 0;
  const size_t chunk = kH2dChunk;
  if (h2d_chunk_ != chunk) {
    for (int i = 0; i < 2; ++i) {
      if (h2d_busy_[i]) { int rc = wait_event(h2d_ev_[i], "staging"); if (rc) return rc; h2d_busy_[i] = false; }
      if (h2d_buf_[i]) pin_release(h2d_buf_[i], h2d_chunk_);
      h2d_buf_[i] = nullptr;
    }
    h2d_chunk_ = chunk;
  }
  for (int i = 0; i < 2; ++i) {
    if (!h2d_buf_[i]) HIPCHK(pin_alloc(&h2d_buf_[i], chunk), "hipHostMalloc(staging)");
    if (!h2d_ev_[i]) HIPCHK(hipEventCreateWithFlags(&h2d_ev_[i], hipEventDisableTiming), "hipEventCreate");
    if (h2d_busy_[i]) { int rc = wait_event(h2d_ev_[i], "staging"); if (rc) return rc; h2d_busy_[i] = false; }
  }
  char* dst = reinterpret_cast<char*>(out_host);
  const char* src = reinterpret_cast<const char*>(src_dev);
  const size_t nchunk = (bytes + chunk - 1) / chunk;
  auto issue = [&](size_t k) -> int {
    const size_t off = k * chunk, len = std::min(chunk, bytes - off);
    HIPCHK(hipMemcpyAsync(h2d_buf_[k & 1], src + off, len, hipMemcpyDeviceToHost, stream_), "L D2H");
    HIPCHK(hipEventRecord(h2d_ev_[k & 1], stream_), "event");
    return 0;
  };
  int rc = issue(0);
  if (rc) return rc;
  for (size_t k = 0; k < nchunk; ++k) {
    if (k + 1 < nchunk && (rc = issue(k + 1))) return rc;
    if ((rc = wait_event(h2d_ev_[k & 1], "L D2H staging"))) return rc;
    const size_t off = k * chunk, len = std::min(chunk, bytes - off);
    std::memcpy(dst + off, h2d_buf_[k & 1], len);
  }
  return 0;
}

int Engine::download(double* out, int64_t count) {
  if (status_) return status_;
  if (count > S_->arena) count = S_->arena;
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  if (loc_off_.empty()) return staged_d2h(out, d_L_, sizeof(double) * (size_t)count);
  // a rank's arena is packed: back into the global layout (zero where nothing is held here)
  std::vector<double> tmp((size_t)std::max<int64_t>(1, arena_elems_));
  {
    int rc = staged_d2h(tmp.data(), d_L_, sizeof(double) * (size_t)arena_elems_);
    if (rc) return rc;
  }
  std::memset(out, 0, sizeof(double) * (size_t)count);
  for (int b = 0; b < S_->nbcol(); ++b) {
    if (loc_off_[(size_t)b] < 0) continue;
    const int64_t off = S_->bcols[b].off, cnt = (int64_t)S_->bcols[b].nrow * S_->bcols[b].width;
    if (off >= count) continue;
    std::memcpy(out + off, tmp.data() + loc_off_[(size_t)b], sizeof(double) * (size_t)std::min(cnt, count - off));
  }
  return 0;
}

int Engine::prepare_solve() {
  if (solve_ready_) return 0;
  const Symbolic& S = *S_;
  build_solve_program(S, prog_.pw, prog_.cb, sprog_, opt_.nranks > 1 ? owner_.data() : nullptr, opt_.rank);
  if (!loc_off_.empty())
    for (size_t b = 0; b < sprog_.units.size(); ++b)   // (units of block columns not held here are never launched)
      if (loc_off_[b] >= 0) sprog_.units[b].off = loc_off_[b];
  {
    TableStager tab;
    tab.add(&d_sunits_, sprog_.units);
    tab.add(&d_slist_, sprog_.diag_list);
    tab.add(&d_stiles_, sprog_.tiles);
    HIPCHK(tab.commit(&d_solve_tables_, [this](void** q, size_t b) { return dalloc(q, b); }), "upload solve tables");
  }
  HIPCHK(dalloc((void**)&d_y_, sizeof(double) * 4 * (size_t)std::max(1, S.n)), "hipMalloc(y)");
  {
    const char* e = std::getenv("SPLLT_SOLVE_DIAG4");
    solve_four_ = prog_.pw == 64 && prog_.cb == 64 && !(e && std::atoi(e) == 0);
  }
  solve_ready_ = true;
  return 0;
}

// Substitution on device vectors in pivot order (y[q * n + p], q < nrhs), in place.
// phase -1: everything that `job` asks for; 0/1/2: the three phases of a
// partitioned solve (schedule.hpp, SolveProgram).
int Engine::solve_dev(double* y_dev, int nrhs, int job, int phase) {
  if (status_) return status_;
  if (job < 0 || job > 2 || phase < -1 || phase > 2 || nrhs < 0 || !y_dev) return -10;
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  int rc = prepare_solve();
  if (rc) return rc;
  const int n = S_->n;
  const bool do_fwd = job == 0 || job == 1, do_bwd = job == 0 || job == 2;
  for (int done = 0; done < nrhs;) {
    const int left = nrhs - done;
    const int cur = left >= 4 ? 4 : (left >= 2 ? 2 : 1);   // kernel variants: 4, 2 or 1 per sweep
    double* y = y_dev + (int64_t)done * n;
    auto run = [&](const std::vector<SolveLaunch>& ls, size_t a, size_t b) {
      for (size_t i = a; i < b; ++i) {
        // block columns of at most four 64-wide panels: the diagonal kernel that reads L in one round trip
        bool four = solve_four_ && (ls[i].kind == SV_DIAG_FWD || ls[i].kind == SV_DIAG_BWD);
        for (int64_t q = ls[i].first; four && q < ls[i].first + ls[i].count; ++q)
          four = sprog_.units[(size_t)sprog_.diag_list[(size_t)q]].w <= 256;
        // a launch on ONE block column (every step of the upper levels): its descriptor by value
        const SolveUnit* one = nullptr;
        if (ls[i].count > 0) {
          if (ls[i].kind == SV_DIAG_FWD || ls[i].kind == SV_DIAG_BWD) {
            if (ls[i].count == 1) one = &sprog_.units[(size_t)sprog_.diag_list[(size_t)ls[i].first]];
          } else {
            const UpdTile* tl = sprog_.tiles.data() + ls[i].first;
            bool same = true;
            for (int64_t q = 0; same && q < ls[i].count; ++q) same = tl[q].unit == tl[0].unit && tl[q].ti == (short)q;
            if (same && ls[i].count < 32768) one = &sprog_.units[(size_t)tl[0].unit];
          }
        }
        launch_solve(stream_, ls[i].kind, d_slist_, d_stiles_, ls[i].first, ls[i].count, d_sunits_, d_L_,
                     d_dinv_, d_rlist_, y, cur, (int64_t)n, four, one);
      }
    };
    const size_t nf = sprog_.fwd.size(), nb = sprog_.bwd.size();
    if (do_fwd && (phase == -1 || phase == 0)) run(sprog_.fwd, 0, sprog_.fwd_nsub);
    if (do_fwd && (phase == -1 || phase == 1)) run(sprog_.fwd, sprog_.fwd_nsub, nf);
    if (do_bwd && (phase == -1 || phase == 1)) run(sprog_.bwd, 0, sprog_.bwd_ntop);
    if (do_bwd && (phase == -1 || phase == 2)) run(sprog_.bwd, sprog_.bwd_ntop, nb);
    done += cur;
  }
  HIPCHK(hipGetLastError(), "solve launch");
  return sync_stream(stream_, "solve sync");
}

int Engine::solve(double* x_host, int nrhs, int job) {
  if (status_) return status_;
  if (job < 0 || job > 2) return -10;
  const Symbolic& S = *S_;
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  int rc = prepare_solve();
  if (rc) return rc;
  const int n = S.n;
  // up to four right-hand sides per sweep: every entry of L is read once for all of them
  std::vector<double> yh((size_t)n * 4);
  for (int done = 0; done < nrhs;) {
    const int left = nrhs - done;
    const int cur = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
    for (int q = 0; q < cur; ++q) {
      const double* xr = x_host + (int64_t)(done + q) * n;
      double* yq = yh.data() + (size_t)q * n;
      for (int i = 0; i < n; ++i) yq[S.order[i]] = xr[i];
    }
    HIPCHK(hipMemcpyAsync(d_y_, yh.data(), sizeof(double) * (size_t)n * cur, hipMemcpyHostToDevice, stream_), "rhs H2D");
    if (comm_ && opt_.nranks > 1) {
      // partitioned solve inside the library (every rank passes the same right-hand sides and gets
      // the same solution): forward substitution on the own subtrees, all-reduce of the vector,
      // the top tree on every rank, backward substitution on the own subtrees, all-reduce
      if (job != 0) return -98;
      Rccl& R = rccl();
      launch_mask(stream_, d_y_, d_owned_, n, cur, (int64_t)n);
      if ((rc = solve_dev(d_y_, cur, 0, 0))) return rc;
      NCCLCHK(R.all_reduce(d_y_, d_y_, (size_t)n * cur, kNcclDouble, kNcclSum, comm_, stream_), "ncclAllReduce(rhs)");
      if ((rc = solve_dev(d_y_, cur, 0, 1))) return rc;
      if ((rc = solve_dev(d_y_, cur, 0, 2))) return rc;
      launch_mask(stream_, d_y_, d_owned_, n, cur, (int64_t)n);
      NCCLCHK(R.all_reduce(d_y_, d_y_, (size_t)n * cur, kNcclDouble, kNcclSum, comm_, stream_), "ncclAllReduce(x)");
      if ((rc = sync_stream(stream_, "solve sync"))) return rc;
    } else {
      rc = solve_dev(d_y_, cur, job, -1);
      if (rc) return rc;
    }
    HIPCHK(hipMemcpy(yh.data(), d_y_, sizeof(double) * (size_t)n * cur, hipMemcpyDeviceToHost), "x D2H");
    for (int q = 0; q < cur; ++q) {
      double* xr = x_host + (int64_t)(done + q) * n;
      const double* yq = yh.data() + (size_t)q * n;
      for (int i = 0; i < n; ++i) xr[i] = yq[S.order[i]];
    }
    done += cur;
  }
  return 0;
}

// Per-launch device time of one factorization from HIP events on the stream each launch
// runs on.  serial = true: the whole program on one stream, launch after launch (the
// kernels alone on the chip).  serial = false: the real multi-stream program; every
// launch is bracketed by two events on ITS stream, recorded behind its dependency waits, so
// the elapsed time is the launch's duration under the contention of whatever runs beside it.
int Engine::profile_launches(const double* val_host, int64_t nnz, std::vector<float>& ms, bool serial) {
  if (status_) return status_;
  if (nnz != S_->nnzA) return -10;
  if (!prog_.exchanges.empty()) return -98;   // single-GPU programs only
  const Symbolic& S = *S_;
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  HIPCHK(hipMemcpy(d_val_, val_host, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice), "val H2D");
  HIPCHK(hipMemsetAsync(d_L_, 0, sizeof(double) * (size_t)S.arena, stream_), "memset arena");
  *h_flag_ = INT_MAX;
  HIPCHK(hipMemcpyAsync(d_flag_, h_flag_, sizeof(int), hipMemcpyHostToDevice, stream_), "flag init");
  launch_scatter_val(stream_, d_L_, d_val_, d_map_dst_, d_map_src_, nmap_);
  size_t nl = prog_.launches.size();
  std::vector<hipEvent_t> ev(2 * nl);
  for (auto& e : ev) HIPCHK(hipEventCreate(&e), "event create");
  for (size_t i = 0; i < nl; ++i) {
    const Launch& l = prog_.launches[i];
    hipStream_t st = serial ? stream_ : streams_[l.stream];
    if (!serial)
      for (int w : l.wait)
        if (w >= 0) HIPCHK(hipStreamWaitEvent(st, dag_events_[w], 0), "stream wait");
    HIPCHK(hipEventRecord(ev[2 * i], st), "event");
    Launch bare = l;                       // waits already issued above; keep the record
    for (int& w : bare.wait) w = -1;
    bare.record = -1;
    enqueue_launch(bare, serial);
    HIPCHK(hipEventRecord(ev[2 * i + 1], st), "event");
    if (!serial && l.record >= 0) HIPCHK(hipEventRecord(dag_events_[l.record], st), "event record");
  }
  if (!serial) {
    if (prog_.final_event >= 0)
      HIPCHK(hipStreamWaitEvent(stream_, dag_events_[prog_.final_event], 0), "final wait");
  }
  HIPCHK(hipStreamSynchronize(stream_), "sync");
  ms.assign(nl, 0.f);
  for (size_t i = 0; i < nl; ++i) hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]);
  for (auto& e : ev) hipEventDestroy(e);
  return 0;
}

// When every event of the real multi-stream program was reached, in ms after the value scatter:
// the program exactly as factor_async submits it, except that its events are timing-enabled ones
// of this call (nothing is added to the streams: the records are the program's own).  t[i] = time
// of the event launch i records (-1: the launch records none); t[nl] = the end of the program.
int Engine::timeline(const double* val_host, int64_t nnz, std::vector<float>& t) {
  if (status_) return status_;
  if (nnz != S_->nnzA) return -10;
  if (!prog_.exchanges.empty()) return -98;   // single-GPU programs only
  const Symbolic& S = *S_;
  HIPCHK(hipSetDevice(device_), "hipSetDevice");
  HIPCHK(hipMemcpy(d_val_, val_host, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice), "val H2D");
  HIPCHK(hipMemsetAsync(d_L_, 0, sizeof(double) * (size_t)S.arena, stream_), "memset arena");
  if (!prog_.panel_units.empty())
    HIPCHK(hipMemsetAsync(d_panel_cnt_, 0, sizeof(int) * 2 * prog_.panel_units.size(), stream_), "memset counters");
  *h_flag_ = INT_MAX;
  HIPCHK(hipMemcpyAsync(d_flag_, h_flag_, sizeof(int), hipMemcpyHostToDevice, stream_), "flag init");
  launch_scatter_val(stream_, d_L_, d_val_, d_map_dst_, d_map_src_, nmap_);
  const size_t nl = prog_.launches.size();
  std::vector<hipEvent_t> timed((size_t)prog_.nevents + 2);
  for (auto& e : timed) HIPCHK(hipEventCreate(&e), "event create");
  std::vector<hipEvent_t> keep;
  keep.swap(dag_events_);
  dag_events_.assign(timed.begin(), timed.begin() + prog_.nevents);
  hipEvent_t e0 = timed[(size_t)prog_.nevents], e1 = timed[(size_t)prog_.nevents + 1];
  int rc = 0;
  hipError_t he = hipEventRecord(e0, stream_);
  for (int s = 0; s < ST_COUNT && he == hipSuccess; ++s)
    if (streams_[s] != stream_) he = hipStreamWaitEvent(streams_[s], e0, 0);
  if (he == hipSuccess) rc = enqueue_range(0, nl);
  if (he == hipSuccess && !rc && prog_.final_event >= 0) he = hipStreamWaitEvent(stream_, dag_events_[prog_.final_event], 0);
  if (he == hipSuccess && !rc) he = hipEventRecord(e1, stream_);
  if (he == hipSuccess && !rc) he = hipStreamSynchronize(stream_);
  if (he == hipSuccess && !rc) {
    t.assign(nl + 1, -1.f);
    for (size_t i = 0; i < nl; ++i) {
      const int r = prog_.launches[i].record;
      if (r >= 0) hipEventElapsedTime(&t[i], e0, dag_events_[(size_t)r]);
    }
    hipEventElapsedTime(&t[nl], e0, e1);
  }
  dag_events_.swap(keep);
  for (auto& e : timed) hipEventDestroy(e);
  if (he != hipSuccess) return fail(kErrHip, "timeline", he);
  return rc;
}

// ---------------------------------------------------------------------------
// Submission under a deadline.  Everything spllt_factor does before it returns -- engine
// creation (stream / event borrowing, allocations, synchronous table upload), the staging of
// val, ~650 launches -- is a sequence of runtime calls that normally take microseconds but that
// nobody can interrupt once one of them sits in the driver (the one hang of round 2 that was not
// in a wait sat here).  They run on ONE helper thread per process; the caller waits for the job
// with the deadline of the other waits.  A job that does not come back marks the runtime as
// wedged: that call and every later one fail with SPLLT_ERROR_HIP and the name of the last step
// reached, instead of blocking the caller forever.  (SPLLT_HIP_TIMEOUT_S=0: no helper thread,
// the calls run inline.)
// ---------------------------------------------------------------------------
namespace {
struct SubmitJob {
  std::function<int()> fn;
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  int rc = 0;
};
struct SubmitWorker {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::shared_ptr<SubmitJob>> queue;
  std::atomic<bool> wedged{false};
  SubmitWorker() {
    std::thread([this] {
      for (;;) {
        std::shared_ptr<SubmitJob> job;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [this] { return !queue.empty(); });
          job = queue.front();
          queue.pop_front();
        }
        const int rc = job->fn();
        {
          std::lock_guard<std::mutex> lk(job->mu);
          job->rc = rc;
          job->done = true;
        }
        job->cv.notify_all();
      }
    }).detach();
  }
};
SubmitWorker& submit_worker() {
  static SubmitWorker* w = new SubmitWorker();   // never destroyed: its thread outlives main()
  return *w;
}
}  // namespace

void mark_runtime_wedged() { g_runtime_wedged.store(true); }
bool runtime_wedged() { return g_runtime_wedged.load(); }
void run_pools_teardown_for_test() { pools_teardown(); }

int run_with_deadline(std::function<int()> fn, std::string* why) {
  // (SPLLT_HIP_SUBMIT_TIMEOUT_S: a deadline of its own -- a wait may legitimately be given a few
  // milliseconds by a caller that polls, engine creation may not)
  static const double limit_s = [] {
    const char* e = std::getenv("SPLLT_HIP_SUBMIT_TIMEOUT_S");
    if (e && *e) return std::atof(e);
    return hip_deadline_s() <= 0 ? 0.0 : 180.0;
  }();
  static const bool inline_env = std::getenv("SPLLT_HIP_SUBMIT_INLINE") != nullptr;
  if (limit_s <= 0 || inline_env) return fn();
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return fn();     // (no device: the engine reports that itself)
  SubmitWorker& w = submit_worker();
  if (w.wedged.load()) {
    if (why) *why = std::string("the HIP runtime did not return from an earlier call (") + last_crumb() + ")";
    return kErrHip;
  }
  auto job = std::make_shared<SubmitJob>();
  job->fn = [fn, dev]() -> int {
    (void)hipSetDevice(dev);                             // the caller's device, not the helper thread's default
    return fn();
  };
  {
    std::lock_guard<std::mutex> lk(w.mu);
    w.queue.push_back(job);
  }
  w.cv.notify_one();
  std::unique_lock<std::mutex> lk(job->mu);
  if (job->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return job->done; })) return job->rc;
  w.wedged.store(true);
  mark_runtime_wedged();
  if (why)
    *why = std::string("submission did not return within ") + std::to_string((int)limit_s) + " s; last step: " +
           last_crumb();
  std::fprintf(stderr, "spllt-hip: %s\n", why ? why->c_str() : "submission deadline");
  return kErrHip;
}

}  // namespace spx
