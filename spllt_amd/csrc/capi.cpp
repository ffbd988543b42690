// extern "C" boundary of libspllt_hip.so: the SpLLT C-ABI (include/spllt_iface.h)
// plus the extensions of include/spllt_hip.h.  Mirrors the behaviour of
// reference interfaces/C/spllt_data_ciface.F90: never aborts, messages on
// stderr, status in info->flag.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "engine.hpp"
#include "kernels.hpp"
#include "spllt_hip.h"
#include "symbolic.hpp"

using namespace spx;

namespace {

struct Akeep {
  std::shared_ptr<Symbolic> S;
  SymOptions so;
};

struct Fkeep {
  std::shared_ptr<Symbolic> S;
  std::unique_ptr<Engine> eng;
  EngineOptions eo;
  std::vector<double> hostL;
  bool hostL_valid = false;
  int last_flag = 0;
  std::string last_error;
  // solve memory handed over by spllt_set_mem_solve (borrowed, unused by the host solve)
  double* y = nullptr;
  double* workspace = nullptr;
  long worksize = 0;
  double* xbuf = nullptr;  // multi-GPU exchange buffer (caller-owned device memory)
  bool dead = false;       // a submission never returned: the engine belongs to the stuck helper thread
};

std::mutex g_mu;
std::vector<Fkeep*> g_pending;  // factorizations submitted and not yet waited for

void clear_info(spllt_inform_t* info) {
  if (!info) return;
  std::memset(info, 0, sizeof(*info));
}

void fill_info(const Symbolic& S, spllt_inform_t* info) {
  if (!info) return;
  info->maxdepth = S.maxdepth;
  info->num_factor = (int)S.nnzL;  // truncated like the reference (ciface:77-78)
  info->num_flops = (int)S.flops;
  info->num_nodes = S.nnodes;
  info->stat = 0;
}

int do_wait(Fkeep* f) {
  if (f->dead) return SPLLT_ERROR_HIP;    // (its engine belongs to a submission that never returned)
  if (!f->eng) return f->last_flag;
  if (f->eng->pending()) {
    int rc = f->eng->wait();
    f->last_flag = rc;
    f->hostL_valid = false;
    if (rc == SPLLT_ERROR_NOT_POSDEF) {
      char buf[160];
      if (f->eng->not_posdef_column() < 0)
        std::snprintf(buf, sizeof buf, "matrix is not positive definite (reported by another rank of the partition)");
      else
        std::snprintf(buf, sizeof buf, "matrix is not positive definite (pivot column %d in elimination order)",
                      f->eng->not_posdef_column() + 1);
      f->last_error = buf;
      std::fprintf(stderr, "spllt-hip: %s\n", buf);
    } else if (rc) {
      f->last_error = f->eng->error();
    }
  }
  return f->last_flag;
}

int ensure_hostL(Fkeep* f) {
  do_wait(f);
  if (f->last_flag) return f->last_flag;
  if (!f->eng) return SPLLT_ERROR_PARAMETER;
  if (!f->hostL_valid) {
    f->hostL.resize((size_t)f->S->arena);
    int rc = f->eng->download(f->hostL.data(), f->S->arena);
    if (rc) return rc;
    f->hostL_valid = true;
  }
  return 0;
}

// the SSIDS-style quintuple of spllt_hip_analyse_symbolic (1-based, as SSIDS delivers it)
struct SymbolicIn {
  int nnodes;
  const int* sptr;
  const int* sparent;
  const int64_t* rptr;
  const int* rlist;
};

void analyse_impl(void** akeep, void** fkeep, spllt_options_t* options, int n, const int* ptr,
                  const int* row, spllt_inform_t* info, int* order, const int* order_in,
                  const SymbolicIn* sym = nullptr) {
  clear_info(info);
  if (!akeep || !fkeep || !options || !ptr || !row || n < 0) {
    std::fprintf(stderr, "spllt-hip: spllt_analyse: invalid argument\n");
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  if (options->nb > 1024) {
    // the substitution kernels keep one block column's worth of the right-hand side in LDS
    std::fprintf(stderr, "spllt-hip: spllt_analyse: nb = %d is not supported (nb <= 1024)\n", options->nb);
    if (info) info->flag = SPLLT_ERROR_UNIMPLEMENTED;
    return;
  }
  Akeep* a = static_cast<Akeep*>(*akeep);
  Fkeep* f = static_cast<Fkeep*>(*fkeep);
  if (!a) { a = new (std::nothrow) Akeep(); *akeep = a; }
  if (!f) {
    f = new (std::nothrow) Fkeep();
    *fkeep = f;
    // experiment knob: default chain block of every new handle (spllt_hip_set_chain_block overrides)
    if (f)
      if (const char* e = std::getenv("SPLLT_CHAIN_BLOCK")) f->eo.cb = std::max(1, std::atoi(e));
  }
  if (!a || !f) { if (info) info->flag = SPLLT_ERROR_ALLOCATION; return; }
  a->so.nb = options->nb;
  a->so.nemin = options->nemin;
  a->so.prune_tree = options->prune_tree != 0;
  a->so.ncpu = options->ncpu;
  if (const char* e = std::getenv("SPLLT_HIP_RELAX")) a->so.relax = std::atof(e);  // experiment knob
  // The reference hands ptr/row to SSIDS unchecked (ssids_analyse(check = .false.),
  // src/spllt_analyse_mod.F90:129): a malformed pattern is undefined behaviour there.
  // Here it is a parameter error: column pointers must start at 1 and not decrease,
  // rows must lie in the lower triangle (col <= row <= n) without duplicates.
  {
    bool ok = ptr[0] == 1;
    for (int j = 0; ok && j < n; ++j) ok = ptr[j + 1] >= ptr[j];
    std::vector<int> mark(ok ? (size_t)n : 0, -1);
    for (int j = 0; ok && j < n; ++j)
      for (int64_t e = (int64_t)ptr[j] - 1; ok && e < (int64_t)ptr[j + 1] - 1; ++e) {
        const int r = row[e] - 1;
        ok = r >= j && r < n && mark[r] != j;
        if (ok) mark[r] = j;
      }
    if (!ok) {
      std::fprintf(stderr, "spllt-hip: spllt_analyse: ptr/row is not a valid lower-triangular CSC pattern\n");
      if (info) info->flag = SPLLT_ERROR_PARAMETER;
      return;
    }
  }
  // 1-based int CSC -> 0-based
  std::vector<int64_t> p0((size_t)n + 1);
  for (int j = 0; j <= n; ++j) p0[j] = (int64_t)ptr[j] - 1;
  const int64_t nz = n > 0 ? p0[n] : 0;
  std::vector<int> r0((size_t)std::max<int64_t>(1, nz));
  for (int64_t e = 0; e < nz; ++e) r0[e] = row[e] - 1;
  std::vector<int> uo;
  if (order_in) {
    uo.resize(n);
    for (int i = 0; i < n; ++i) uo[i] = order_in[i] - 1;
  }
  auto S = std::make_shared<Symbolic>();
  int rc;
  try {
    if (sym) {
      // 1-based -> 0-based; the virtual root nnodes+1 becomes nnodes
      const int nn = sym->nnodes;
      if (nn < 1 || !order_in || !sym->sptr || !sym->sparent || !sym->rptr || !sym->rlist || sym->rptr[0] != 1 ||
          sym->rptr[nn] < 1) {
        rc = SPLLT_ERROR_PARAMETER;
      } else {
        std::vector<int> sp(nn + 1), spar(nn), rl((size_t)(sym->rptr[nn] - 1));
        std::vector<int64_t> rp(nn + 1);
        for (int s = 0; s <= nn; ++s) { sp[s] = sym->sptr[s] - 1; rp[s] = sym->rptr[s] - 1; }
        for (int s = 0; s < nn; ++s) spar[s] = sym->sparent[s] - 1;
        for (size_t k = 0; k < rl.size(); ++k) rl[k] = sym->rlist[k] - 1;
        rc = analyse_symbolic(n, p0.data(), r0.data(), nn, sp.data(), spar.data(), rp.data(), rl.data(),
                              uo.data(), a->so, *S);
      }
    } else {
      rc = analyse(n, p0.data(), r0.data(), order_in ? uo.data() : nullptr, a->so, *S);
    }
  } catch (const std::bad_alloc&) {
    rc = SPLLT_ERROR_ALLOCATION;
  }
  if (rc) {
    std::fprintf(stderr, "spllt-hip: spllt_analyse failed with flag %d\n", rc);
    if (info) info->flag = rc;
    return;
  }
  a->S = S;
  f->S = S;
  f->eng.reset();
  f->hostL_valid = false;
  f->last_flag = 0;
  if (order)
    for (int i = 0; i < n; ++i) order[i] = S->order[i] + 1;
  fill_info(*S, info);
}

void factor_impl(void* akeep, void* fkeep, int nnz, const double* val, bool dev, spllt_inform_t* info) {
  clear_info(info);
  Akeep* a = static_cast<Akeep*>(akeep);
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!a || !f || !f->S || !val) {
    std::fprintf(stderr, "spllt-hip: spllt_factor: akeep/fkeep/val provided by the user is empty\n");
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  if ((int64_t)nnz != f->S->nnzA) {
    std::fprintf(stderr, "spllt-hip: spllt_factor: nnz = %d does not match the analysed pattern (%lld)\n", nnz,
                 (long long)f->S->nnzA);
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  if (f->dead) {          // an earlier submission of this handle never returned (below)
    if (info) info->flag = SPLLT_ERROR_HIP;
    return;
  }
  if (f->eng && f->eng->pending()) do_wait(f);
  // engine creation and submission on the helper thread, under the deadline (engine.cpp)
  std::string why;
  int rc = run_with_deadline([f, dev, val, nnz]() -> int {
    if (!f->eng) {
      f->eng.reset(new (std::nothrow) Engine(f->S, f->eo));
      if (!f->eng) return SPLLT_ERROR_ALLOCATION;
      f->eng->set_exchange_buffer(f->xbuf);
    }
    if (f->eng->status()) return f->eng->status();
    return dev ? f->eng->factor_async_dev(val, nnz) : f->eng->factor_async(val, nnz);
  }, &why);
  if (!why.empty()) {
    // the helper thread may still be inside the engine: the handle is dead, its engine is never
    // touched again (spllt_deallocate_fkeep leaks it)
    f->dead = true;
    f->last_flag = SPLLT_ERROR_HIP;
    f->last_error = why;
    if (info) info->flag = SPLLT_ERROR_HIP;
    return;
  }
  if (rc == SPLLT_ERROR_ALLOCATION && !f->eng) { if (info) info->flag = rc; return; }
  f->last_flag = rc;
  f->hostL_valid = false;
  if (rc == 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (std::find(g_pending.begin(), g_pending.end(), f) == g_pending.end()) g_pending.push_back(f);
  } else {
    f->last_error = f->eng->error();
  }
  fill_info(*f->S, info);
  if (info) info->flag = rc;
}

template <class Tp>
static int64_t copy_out(const std::vector<Tp>& v, void* buf, int64_t cap) {
  if (buf) {
    int64_t k = std::min<int64_t>(cap, (int64_t)v.size());
    if (k > 0) std::memcpy(buf, v.data(), sizeof(Tp) * (size_t)k);
  }
  return (int64_t)v.size();
}

}  // namespace

extern "C" {

void spllt_analyse(void** akeep, void** fkeep, spllt_options_t* options, int n, int* ptr, int* row,
                   spllt_inform_t* info, int* order) {
  analyse_impl(akeep, fkeep, options, n, ptr, row, info, order, nullptr);
}

void spllt_hip_analyse_ordered(void** akeep, void** fkeep, spllt_options_t* options, int n,
                               const int* ptr, const int* row, spllt_inform_t* info, int* order,
                               const int* order_in) {
  analyse_impl(akeep, fkeep, options, n, ptr, row, info, order, order_in);
}

void spllt_hip_analyse_symbolic(void** akeep, void** fkeep, spllt_options_t* options, int n,
                                const int* ptr, const int* row, spllt_inform_t* info, int nnodes,
                                const int* sptr, const int* sparent, const int64_t* rptr,
                                const int* rlist, const int* order_in) {
  SymbolicIn sym{nnodes, sptr, sparent, rptr, rlist};
  std::vector<int> order_out((size_t)std::max(n, 1));
  analyse_impl(akeep, fkeep, options, n, ptr, row, info, order_out.data(), order_in, &sym);
}

void spllt_factor(void* akeep, void* fkeep, spllt_options_t* options, int nnz, double* val,
                  spllt_inform_t* info) {
  (void)options;
  factor_impl(akeep, fkeep, nnz, val, false, info);
}

void spllt_hip_factor_dev(void* akeep, void* fkeep, spllt_options_t* options, int nnz,
                          const double* val_dev, spllt_inform_t* info) {
  (void)options;
  factor_impl(akeep, fkeep, nnz, val_dev, true, info);
}

void spllt_wait(void) {
  std::vector<Fkeep*> todo;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    todo.swap(g_pending);
  }
  for (Fkeep* f : todo) do_wait(f);
}

int spllt_hip_wait(void* fkeep) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f) return SPLLT_ERROR_PARAMETER;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.erase(std::remove(g_pending.begin(), g_pending.end(), f), g_pending.end());
  }
  return do_wait(f);
}

void spllt_solve_workspace_size(void* fkeep, int nworker, int nrhs, long* size) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!size) return;
  *size = 0;
  if (!f || !f->S) return;
  if (nworker < 1) nworker = 1;
  // reference formula, src/spllt_data_mod.F90:655
  *size = (long)f->S->n * nrhs + ((long)f->S->maxmn + f->S->n) * nrhs * nworker;
}

void spllt_prepare_solve(void* akeep, void* fkeep, int nb, int nrhs, long* worksize,
                         spllt_inform_t* info) {
  (void)akeep; (void)nb;
  clear_info(info);
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S) {
    std::fprintf(stderr, "spllt-hip: Error, fkeep provided by the user is empty\n");
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  spllt_solve_workspace_size(fkeep, 1, nrhs, worksize);
  fill_info(*f->S, info);
}

void spllt_set_mem_solve(void* akeep, void* fkeep, int nb, int nrhs, long worksize, double* y,
                         double* workspace, spllt_inform_t* info) {
  (void)akeep; (void)nb; (void)nrhs;
  clear_info(info);
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S) {
    std::fprintf(stderr, "spllt-hip: Error, fkeep provided by the user is empty\n");
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  if (!y) std::fprintf(stderr, "spllt-hip: Error, y provided by the user is empty\n");
  if (!workspace) std::fprintf(stderr, "spllt-hip: Error, workspace provided by the user is empty\n");
  f->y = y;
  f->workspace = workspace;
  f->worksize = worksize;
  fill_info(*f->S, info);
}

void spllt_solve(void* fkeep, spllt_options_t* options, int* order, int nrhs, double* x,
                 spllt_inform_t* info, int job) {
  (void)options; (void)order;  // `order` is ignored by the reference too (ciface:404-419)
  clear_info(info);
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !x) {
    std::fprintf(stderr, "spllt-hip: Error, fkeep/x provided by the user is empty\n");
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  if (job < 0 || job > 2) {
    // reference src/spllt_solve_mod.F90:216-220
    std::fprintf(stderr, "Unknown requested job = %2d returned code : %4d\n", job, SPLLT_ERROR_PARAMETER);
    if (info) info->flag = SPLLT_ERROR_PARAMETER;
    return;
  }
  // Solve on the device-resident factor (no D2H of L).
  int rc = do_wait(f);
  if (rc == 0 && !f->eng) rc = SPLLT_ERROR_PARAMETER;  // nothing factorized yet
  if (rc) { if (info) info->flag = rc; return; }
  if (f->eo.nranks > 1 && !f->eng->has_communicator()) {
    // (with spllt_hip_set_communicator the engine runs the two all-reduces itself, below)
    // A partitioned factor is spread over the ranks (own subtrees + replicated top
    // tree); this process holds only its part, so a local substitution would be
    // wrong.  The partitioned solve is spllt_hip_solve_dev in three phases with the
    // caller's all-reduce in between (spllt_amd/multigpu.py, DistributedFactorization.solve).
    std::fprintf(stderr, "spllt-hip: spllt_solve on a partitioned (multi-GPU) factor needs the caller's "
                         "exchange: use spllt_hip_solve_dev (phases 0, 1, 2)\n");
    if (info) info->flag = SPLLT_ERROR_UNIMPLEMENTED;
    return;
  }
  rc = f->eng->solve(x, nrhs, job);
  if (rc) { if (info) info->flag = rc; return; }
  fill_info(*f->S, info);
}

int spllt_hip_solve_dev(void* fkeep, void* y_dev, int nrhs, int job, int phase) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !y_dev) return SPLLT_ERROR_PARAMETER;
  int rc = do_wait(f);
  if (rc == 0 && !f->eng) rc = SPLLT_ERROR_PARAMETER;  // nothing factorized yet
  if (rc) return rc;
  return f->eng->solve_dev(static_cast<double*>(y_dev), nrhs, job, phase);
}

void spllt_solve_worker(void* fkeep, spllt_options_t* options, int* order, int nrhs, double* x,
                        spllt_inform_t* info, int job, double* workspace, long worksize, void* tm) {
  (void)workspace; (void)worksize; (void)tm;
  spllt_solve(fkeep, options, order, nrhs, x, info, job);
}

// reference src/utils_mod.F90:432-478 (check_backward_error_multi): scaled
// backward error ||b - A x||_2 / (||b||_2 + max|a_ij| ||x||_2), pass <= 1e-14.
void spllt_chkerr(int n, int* ptr, int* row, double* val, int nrhs, double* x, double* rhs) {
  if (!ptr || !row || !val || !x || !rhs) {
    std::fprintf(stderr, "spllt-hip: Error, an array provided by the user is empty\n");
    return;
  }
  double amax = 0;
  for (int e = 0; e < ptr[n] - 1; ++e) amax = std::max(amax, std::fabs(val[e]));
  int ok = 0;
  std::vector<double> res(n);
  for (int r = 0; r < nrhs; ++r) {
    const double* xr = x + (int64_t)r * n;
    const double* br = rhs + (int64_t)r * n;
    for (int i = 0; i < n; ++i) res[i] = br[i];
    for (int j = 0; j < n; ++j)
      for (int e = ptr[j] - 1; e < ptr[j + 1] - 1; ++e) {
        int i = row[e] - 1;
        res[i] -= val[e] * xr[j];
        if (i != j) res[j] -= val[e] * xr[i];
      }
    double nr = 0, nb = 0, nx = 0;
    for (int i = 0; i < n; ++i) { nr += res[i] * res[i]; nb += br[i] * br[i]; nx += xr[i] * xr[i]; }
    double err = std::sqrt(nr) / (std::sqrt(nb) + amax * std::sqrt(nx));
    if (err != err) {
      std::printf("Backward error of rhs %3d is equal to a NAN\n", r + 1);
    } else if (err > 1e-14) {
      std::fprintf(stderr, "Wrong Bwd error for %4d/%4d : %10.2e\n", r + 1, nrhs, err);
    } else {
      std::fprintf(stderr, "Bwd error for %4d/%4d : %10.2e\n", r + 1, nrhs, err);
      ok++;
    }
  }
  std::fprintf(stderr, "Backward error... ok for %3d/%3d\n", ok, nrhs);
}

void spllt_deallocate_fkeep(void** fkeep, int* stat) {
  if (stat) *stat = 0;
  if (!fkeep || !*fkeep) return;
  Fkeep* f = static_cast<Fkeep*>(*fkeep);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.erase(std::remove(g_pending.begin(), g_pending.end(), f), g_pending.end());
  }
  if (f->dead) {
    // a submission of this handle never returned: the helper thread may still be inside the
    // engine, and if it was merely slow it goes on to write f->eng and to read the symbolic
    // structure and the staged values through f -- the whole handle is leaked, not just its parts
    // (and `val` of the spllt_factor call that failed must stay valid: spllt_iface.h)
    *fkeep = nullptr;
    return;
  }
  delete f;
  *fkeep = nullptr;
}

void spllt_deallocate_akeep(void** akeep, int* stat) {
  if (stat) *stat = 0;
  if (!akeep || !*akeep) return;
  delete static_cast<Akeep*>(*akeep);
  *akeep = nullptr;
}

// The reference's task manager drives the OpenMP solve tasks; the stream-DAG
// engine needs none.  A token object keeps init/deallocate pairs well-formed.
void spllt_task_manager_init(void** task_manager) {
  if (task_manager) *task_manager = new int(0);
}
void spllt_task_manager_deallocate(void** task_manager, int* stat) {
  if (stat) *stat = 0;
  if (!task_manager || !*task_manager) return;
  delete static_cast<int*>(*task_manager);
  *task_manager = nullptr;
}

void spllt_all(void** akeep, void** fkeep, spllt_options_t* options, int n, int nnz, int nrhs,
               int nb, int* ptr, int* row, double* val, double* x, double* rhs,
               spllt_inform_t* info) {
  if (options) options->nb = nb;
  std::vector<int> order((size_t)std::max(1, n));
  spllt_analyse(akeep, fkeep, options, n, ptr, row, info, order.data());
  if (info && info->flag < 0) return;
  spllt_factor(*akeep, *fkeep, options, nnz, val, info);
  if (info && info->flag < 0) return;
  int rc = spllt_hip_wait(*fkeep);
  if (rc) { if (info) info->flag = rc; return; }
  long ws = 0;
  spllt_prepare_solve(*akeep, *fkeep, nb, nrhs, &ws, info);
  if (x != rhs) std::memcpy(x, rhs, sizeof(double) * (size_t)n * nrhs);
  spllt_solve(*fkeep, options, order.data(), nrhs, x, info, 0);
  if (info && info->flag < 0) return;
  spllt_chkerr(n, ptr, row, val, nrhs, x, rhs);
}

// ---------------------------------------------------------------------------
// extensions
// ---------------------------------------------------------------------------
int spllt_hip_sym_info(const void* akeep, spllt_hip_sym_info_t* out) {
  const Akeep* a = static_cast<const Akeep*>(akeep);
  if (!a || !a->S || !out) return SPLLT_ERROR_PARAMETER;
  const Symbolic& S = *a->S;
  std::memset(out, 0, sizeof(*out));
  out->n = S.n; out->nnz_a = S.nnzA; out->nnodes = S.nnodes; out->nbcol = S.nbcol();
  out->nblk = S.nblk; out->arena = S.arena; out->nnz_l = S.nnzL; out->flops = S.flops;
  out->rlist_len = (int64_t)S.rlist.size();
  out->nb = S.nb; out->maxmn = S.maxmn; out->maxdepth = S.maxdepth;
  int nl = 0;
  for (int s = 0; s < S.nnodes; ++s) nl = std::max(nl, S.level[s] + 1);
  out->nlevels = nl;
  std::snprintf(out->ordering, sizeof out->ordering, "%s", S.ordering.c_str());
  return 0;
}

int64_t spllt_hip_sym_get(const void* akeep, const char* name, void* buf, int64_t cap) {
  const Akeep* a = static_cast<const Akeep*>(akeep);
  if (!a || !a->S || !name) return -1;
  const Symbolic& S = *a->S;
  std::string k(name);
  if (k == "order") return copy_out(S.order, buf, cap);
  if (k == "sptr") return copy_out(S.sptr, buf, cap);
  if (k == "sparent") return copy_out(S.sparent, buf, cap);
  if (k == "rlist") return copy_out(S.rlist, buf, cap);
  if (k == "small") return copy_out(S.small, buf, cap);
  if (k == "level") return copy_out(S.level, buf, cap);
  if (k == "rptr") return copy_out(S.rptr, buf, cap);
  if (k == "map_dst") return copy_out(S.map_dst, buf, cap);
  if (k == "map_src") return copy_out(S.map_src, buf, cap);
  if (k == "lmap_ptr") return copy_out(S.lmap_ptr, buf, cap);
  if (k == "weight") return copy_out(S.weight, buf, cap);
  if (k == "node_bcol0") return copy_out(S.node_bcol0, buf, cap);
  if (k.rfind("bcol_", 0) == 0) {
    const int nb = S.nbcol();
    if (k == "bcol_off") {
      std::vector<int64_t> v(nb);
      for (int b = 0; b < nb; ++b) v[b] = S.bcols[b].off;
      return copy_out(v, buf, cap);
    }
    std::vector<int> v(nb);
    for (int b = 0; b < nb; ++b) {
      const BlockCol& B = S.bcols[b];
      v[b] = k == "bcol_node" ? B.node : k == "bcol_width" ? B.width : k == "bcol_r0" ? B.r0 : k == "bcol_nrow" ? B.nrow : -1;
    }
    return copy_out(v, buf, cap);
  }
  return -1;
}

int spllt_hip_set_engine(void* fkeep, int panel_width, int tile, int flags) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f) return SPLLT_ERROR_PARAMETER;
  if (f->eng) return SPLLT_ERROR_PARAMETER;  // too late
  if (panel_width > 0) f->eo.pw = std::min(panel_width, kPanelMax);
  if (tile > 0) f->eo.tile = tile;
  f->eo.lookahead = (flags & 2) == 0;       // bit 1 set: single-stream program
  f->eo.slice_between = (flags & 64) == 0;  // bit 6 set: inter-node updates only at the end of a level
  f->eo.poison_lds = (flags & 128) != 0;    // bit 7 set: debug, LDS poisoned before every launch
  if (flags & 256) f->eo.reserve_cus = 0;   // bit 8 set: no CU reservation for the chain
  if (flags & 1024) f->eo.zones = 1;        // bit 10 / 11: force the zone pipeline (and the atomic
  if (flags & 2048) f->eo.zones = 0;        // trailing updates that go with it) on / off
  f->eo.fused_panel = (flags & 512) == 0;   // bit 9 set: no fused panel launches (POTRF, TRSM, update apart)
  f->eo.deterministic = (flags & 4096) != 0;
  if (flags & 8192) f->eo.dist_top = 1;     // bit 13 / 14: top tree of a partitioned factorization
  if (flags & 16384) f->eo.dist_top = 0;    // distributed over the ranks / replicated on every rank  // bit 12: no atomics (buffer + ordered gather)
  if (flags & 32768) f->eo.graph = 1;       // bit 15 / 16: HIP-graph replay, one chain in program order /
  if (flags & 65536) f->eo.graph = 2;       // the DAG of the multi-stream program
  if (flags & 131072) f->eo.graph = 0;      // bit 17: eager launches
  if (flags & 262144) f->eo.subtrees = 1;   // bit 18 / 19: small subtrees as single device tasks (L_SUBTREE)
  if (flags & 524288) f->eo.subtrees = 0;   // on / off
  return 0;
}

// test hooks of the process-wide "runtime is wedged" state (engine.cpp): "wedge" sets it, "wedged"
// reads it, "teardown" runs the atexit handler of the pools now
int spllt_hip_debug(const char* what) {
  if (!what) return -1;
  const std::string w(what);
  if (w == "wedge") { mark_runtime_wedged(); return 0; }
  if (w == "wedged") return runtime_wedged() ? 1 : 0;
  if (w == "teardown") { run_pools_teardown_for_test(); return 0; }
  return -1;
}

int spllt_hip_set_chain_block(void* fkeep, int chain_block) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || chain_block < 1) return SPLLT_ERROR_PARAMETER;
  if (f->eng) return SPLLT_ERROR_PARAMETER;  // too late
  f->eo.cb = chain_block;
  return 0;
}

// The program of a handle that has no engine (yet): built on the host, no GPU needed.
static void build_local_program(Fkeep* f, Program& local) {
  EngineOptions eo = f->eo;                 // (the handle's options stay unresolved)
  ScheduleOptions so = schedule_options(*f->S, eo);
  std::vector<int> owner, top_owner;
  partition_options(*f->S, f->eo, owner, top_owner, so);
  build_program(*f->S, so, local);
}

// ---- multi-GPU partition ---------------------------------------------------
// Works without a GPU (tests inspect the partition and the two-phase program):
// the owners are recomputed from the symbolic structure when no engine exists.
static void partition_tables(Fkeep* f, std::vector<int>& owner, std::vector<int>& top,
                             std::vector<char>& keep, int64_t& elems) {
  const Symbolic& S = *f->S;
  assign_owners(S, f->eo.nranks, owner);
  top.clear();
  elems = 0;
  for (int b = 0; b < S.nbcol(); ++b)
    if (owner[S.bcols[b].node] < 0) {
      top.push_back(b);
      elems += (int64_t)S.bcols[b].nrow * S.bcols[b].width;
    }
  keep.assign(S.map_dst.size(), 0);
  for (int b = 0; b < S.nbcol(); ++b) {
    const int own = owner[S.bcols[b].node];
    if ((own == f->eo.rank) || (own < 0 && f->eo.rank == 0))
      for (int64_t i = S.lmap_ptr[b]; i < S.lmap_ptr[b + 1]; ++i) keep[i] = 1;
  }
}

int spllt_hip_set_partition(void* fkeep, int rank, int nranks, int64_t* exchange_elems) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || nranks < 1 || rank < 0 || rank >= nranks) return SPLLT_ERROR_PARAMETER;
  if (f->eng) return SPLLT_ERROR_PARAMETER;  // too late
  f->eo.rank = rank;
  f->eo.nranks = nranks;
  if (exchange_elems) {
    *exchange_elems = 0;
    if (nranks > 1) {   // the largest exchange of this rank's program
      Program local;
      build_local_program(f, local);
      *exchange_elems = local.xbuf_elems;
    }
  }
  return 0;
}

void* spllt_hip_engine_stream(void* fkeep) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S) return nullptr;
  if (!f->eng) {   // created here so that the caller can order its collective on it before the first factor
    f->eng.reset(new (std::nothrow) Engine(f->S, f->eo));
    if (!f->eng || f->eng->status()) return nullptr;
    f->eng->set_exchange_buffer(f->xbuf);
  }
  return (void*)f->eng->stream();
}

void* spllt_hip_exchange_stream(void* fkeep) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !f->eng) return spllt_hip_engine_stream(fkeep);
  return (void*)f->eng->pending_exchange_stream();
}

int spllt_hip_set_communicator(void* fkeep, void* nccl_comm) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S) return SPLLT_ERROR_PARAMETER;
  if (f->dead) return SPLLT_ERROR_HIP;
  if (!f->eng) {
    f->eng.reset(new (std::nothrow) Engine(f->S, f->eo));
    if (!f->eng) return SPLLT_ERROR_ALLOCATION;
    f->eng->set_exchange_buffer(f->xbuf);
  }
  if (f->eng->status()) { f->last_error = f->eng->error(); return f->eng->status(); }
  int rc = f->eng->set_communicator(nccl_comm);
  if (rc) f->last_error = f->eng->error();
  return rc;
}

int spllt_hip_set_exchange_buffer(void* fkeep, void* dev_ptr) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S) return SPLLT_ERROR_PARAMETER;
  f->xbuf = static_cast<double*>(dev_ptr);
  if (f->eng) f->eng->set_exchange_buffer(f->xbuf);
  return 0;
}

int spllt_hip_pending_exchange(void* fkeep) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->eng) return -1;
  return f->eng->pending_exchange();
}

int spllt_hip_continue(void* fkeep) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->eng) return SPLLT_ERROR_PARAMETER;
  int rc = f->eng->continue_after_exchange();
  f->last_flag = rc;
  if (rc == 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (std::find(g_pending.begin(), g_pending.end(), f) == g_pending.end()) g_pending.push_back(f);
  }
  return rc;
}

int64_t spllt_hip_partition_get(void* fkeep, const char* name, void* buf, int64_t cap) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !name) return -1;
  std::vector<int> owner, top;
  std::vector<char> keep;
  int64_t elems = 0;
  partition_tables(f, owner, top, keep, elems);
  std::string k(name);
  auto raw = [&](const void* p, size_t bytes) -> int64_t {
    if (buf && bytes) std::memcpy(buf, p, std::min<size_t>(bytes, (size_t)cap));
    return (int64_t)bytes;
  };
  if (k == "arena_elems") {   // int64 x 2: doubles of the factor arena held on this rank's device, of the whole arena
    int64_t v[2] = {f->eng ? f->eng->arena_elems() : f->S->arena, f->S->arena};
    return raw(v, sizeof v);
  }
  if (k == "owner") return raw(owner.data(), owner.size() * sizeof(int));
  if (k == "top_bcol_owner") {   // empty: the top tree is replicated
    ScheduleOptions so;
    std::vector<int> o2, top_owner;
    partition_options(*f->S, f->eo, o2, top_owner, so);
    return raw(top_owner.data(), top_owner.size() * sizeof(int));
  }
  if (k == "top_bcols") return raw(top.data(), top.size() * sizeof(int));
  if (k == "map_keep") return raw(keep.data(), keep.size());
  return -1;
}

int spllt_hip_get_factor(void* fkeep, double* out, int64_t count) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !out) return SPLLT_ERROR_PARAMETER;
  int rc = ensure_hostL(f);
  if (rc) return rc;
  std::memcpy(out, f->hostL.data(), sizeof(double) * (size_t)std::min<int64_t>(count, f->S->arena));
  return 0;
}

double* spllt_hip_device_factor(void* fkeep) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  return (f && f->eng) ? f->eng->device_L() : nullptr;
}

int spllt_hip_factor_times(void* fkeep, double* submit_ms, double* device_ms, double* h2d_ms, int* launches) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->eng) return SPLLT_ERROR_PARAMETER;
  const FactorStats& st = f->eng->stats();
  if (submit_ms) *submit_ms = st.submit_ms;
  if (device_ms) *device_ms = st.device_ms;
  if (h2d_ms) *h2d_ms = st.h2d_ms;
  if (launches) *launches = st.launches;
  return 0;
}

int64_t spllt_hip_program_get(void* fkeep, const char* name, void* buf, int64_t cap) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !name) return -1;
  // The program can be inspected without a GPU: build it on demand.
  Program local;
  const Program* P;
  if (f->eng && !f->eng->status()) {
    P = &f->eng->program();
  } else {
    build_local_program(f, local);
    P = &local;
  }
  std::string k(name);
  auto raw = [&](const void* p, size_t bytes) -> int64_t {
    if (buf && bytes) std::memcpy(buf, p, std::min<size_t>(bytes, (size_t)cap));
    return (int64_t)bytes;
  };
  if (k == "launches") {
    std::vector<int64_t> v;
    for (const Launch& l : P->launches) {
      v.push_back(l.kind); v.push_back(l.level); v.push_back(l.first);
      v.push_back(l.count); v.push_back(l.tile); v.push_back((int64_t)l.flops);
      v.push_back(l.stream); v.push_back(l.record);
      for (int w : l.wait) v.push_back(w);
    }
    return raw(v.data(), v.size() * sizeof(int64_t));
  }
  if (k == "potrf") return raw(P->potrf_units.data(), P->potrf_units.size() * sizeof(PotrfUnit));
  if (k == "units") return raw(P->units.data(), P->units.size() * sizeof(UpdUnit));
  if (k == "tiles") return raw(P->tiles.data(), P->tiles.size() * sizeof(UpdTile));
  if (k == "relpos") return raw(P->relpos.data(), P->relpos.size() * sizeof(int));
  if (k == "exchanges") {   // int64 x 5 per exchange: kind, first_item, nitems, elems, chunk
    std::vector<int64_t> t;
    for (const Exchange& e : P->exchanges) {
      t.push_back(e.kind); t.push_back(e.first_item); t.push_back(e.nitems); t.push_back(e.elems); t.push_back(e.chunk);
    }
    return raw(t.data(), t.size() * sizeof(int64_t));
  }
  if (k == "xitems") {      // int64 x 6 per item: block column, root, offset in the buffer, count,
    std::vector<int64_t> t; // offset in the arena / dinv scratch, space (0 arena, 1 dinv)
    for (const ExchangeItem& e : P->xitems) {
      t.push_back(e.bcol); t.push_back(e.root); t.push_back(e.xoff); t.push_back(e.count);
      t.push_back(e.off); t.push_back(e.space);
    }
    return raw(t.data(), t.size() * sizeof(int64_t));
  }
  if (k == "xbuf_elems") return raw(&P->xbuf_elems, sizeof(int64_t));
  if (k == "panels") return raw(P->panel_units.data(), P->panel_units.size() * sizeof(PanelUnit));
  if (k == "sub_tasks") return raw(P->sub_tasks.data(), P->sub_tasks.size() * sizeof(SubTask));
  if (k == "sub_nodes") return raw(P->sub_nodes.data(), P->sub_nodes.size() * sizeof(SubNode));
  if (k == "gen_size") { int64_t v = P->gen_size; return raw(&v, sizeof v); }
  if (k == "chains") return raw(P->chain_units.data(), P->chain_units.size() * sizeof(ChainUnit));
  if (k == "chain_block") { int64_t v = P->cb; return raw(&v, sizeof v); }
  if (k == "panel_width") { int64_t v = P->pw; return raw(&v, sizeof v); }
  if (k == "gather_tiles") return raw(P->gather_tiles.data(), P->gather_tiles.size() * sizeof(GatherTile));
  if (k == "gather_items") return raw(P->gather_items.data(), P->gather_items.size() * sizeof(GatherItem));
  if (k == "scratch_size") { int64_t v = P->scratch_size; return raw(&v, sizeof v); }
  if (k == "dinv_size") { int64_t v = P->dinv_size; return raw(&v, sizeof v); }
  if (k.rfind("solve_", 0) == 0) {
    // the substitution program (partition-aware like the factor program)
    SolveProgram sp;
    std::vector<int> owner;
    if (f->eo.nranks > 1) assign_owners(*f->S, f->eo.nranks, owner);
    build_solve_program(*f->S, f->eo.pw > 0 ? f->eo.pw : kPanelMax, P->cb, sp,
                        f->eo.nranks > 1 ? owner.data() : nullptr, f->eo.rank);
    if (k == "solve_units") return raw(sp.units.data(), sp.units.size() * sizeof(SolveUnit));
    if (k == "solve_list") return raw(sp.diag_list.data(), sp.diag_list.size() * sizeof(int));
    if (k == "solve_tiles") return raw(sp.tiles.data(), sp.tiles.size() * sizeof(UpdTile));
    if (k == "solve_fwd" || k == "solve_bwd") {
      std::vector<int64_t> v;
      for (const SolveLaunch& l : (k == "solve_fwd" ? sp.fwd : sp.bwd)) {
        v.push_back(l.kind); v.push_back(l.level); v.push_back(l.first); v.push_back(l.count);
      }
      return raw(v.data(), v.size() * sizeof(int64_t));
    }
    if (k == "solve_split") {
      int64_t v[2] = {(int64_t)sp.fwd_nsub, (int64_t)sp.bwd_ntop};
      return raw(v, sizeof v);
    }
  }
  return -1;
}

static int profile_impl(void* fkeep, const double* val, int nnz, float* ms, int capacity, bool serial);
int spllt_hip_profile(void* fkeep, const double* val, int nnz, float* ms, int capacity) {
  return profile_impl(fkeep, val, nnz, ms, capacity, true);
}
int spllt_hip_profile_in_program(void* fkeep, const double* val, int nnz, float* ms, int capacity) {
  return profile_impl(fkeep, val, nnz, ms, capacity, false);
}
static int profile_impl(void* fkeep, const double* val, int nnz, float* ms, int capacity, bool serial) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !val) return SPLLT_ERROR_PARAMETER;
  if (f->dead) return SPLLT_ERROR_HIP;
  (void)do_wait(f);            // a factorization still in flight owns the streams and the events
  if (!f->eng) f->eng.reset(new (std::nothrow) Engine(f->S, f->eo));
  if (!f->eng || f->eng->status()) return SPLLT_ERROR_HIP;
  std::vector<float> v;
  int rc = f->eng->profile_launches(val, nnz, v, serial);
  if (rc) return rc;
  for (int i = 0; i < (int)v.size() && i < capacity; ++i) ms[i] = v[i];
  f->hostL_valid = false;
  return (int)v.size();
}

int spllt_hip_timeline(void* fkeep, const double* val, int nnz, float* t_ms, int capacity) {
  Fkeep* f = static_cast<Fkeep*>(fkeep);
  if (!f || !f->S || !val) return SPLLT_ERROR_PARAMETER;
  if (f->dead) return SPLLT_ERROR_HIP;
  (void)do_wait(f);            // a factorization still in flight owns the streams and the events
  if (!f->eng) f->eng.reset(new (std::nothrow) Engine(f->S, f->eo));
  if (!f->eng || f->eng->status()) return SPLLT_ERROR_HIP;
  std::vector<float> v;
  int rc = f->eng->timeline(val, nnz, v);
  if (rc) return rc;
  for (int i = 0; i < (int)v.size() && i < capacity; ++i) t_ms[i] = v[i];
  f->hostL_valid = false;
  return (int)v.size();
}

int spllt_hip_last_flag(const void* fkeep) {
  const Fkeep* f = static_cast<const Fkeep*>(fkeep);
  return f ? (f->dead ? SPLLT_ERROR_HIP : f->last_flag) : SPLLT_ERROR_PARAMETER;
}

const char* spllt_hip_last_error(const void* fkeep) {
  const Fkeep* f = static_cast<const Fkeep*>(fkeep);
  if (!f) return "";
  // (a handle whose communicator is a one-rank stand-in says so for as long as it lives)
  if (f->last_error.empty() && f->eng && f->eng->comm_rehearsal())
    return "rehearsal: a one-rank communicator stands in for the partition's ranks (SPLLT_HIP_COMM_REHEARSAL); "
           "the factor of this handle is not the factor of the matrix";
  return f->last_error.c_str();
}

const char* spllt_hip_version(void) { return "spllt-hip 0.1 (gfx950)"; }

}  // extern "C"
