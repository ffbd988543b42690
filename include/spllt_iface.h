/*
 * spllt_iface.h -- the drop-in C-ABI of libspllt_hip.so.
 *
 * Binary-compatible with the C interface of NLAFET/SpLLT: same symbol names,
 * same struct layouts, same argument order and meaning, so a program built
 * against the reference's include/spllt_iface.h links against this library
 * unchanged.  Each entry point cites the reference declaration it replaces
 * (include/spllt_iface.h:LINE) and the Fortran body behind it
 * (interfaces/C/spllt_data_ciface.F90:LINE).
 *
 * Conventions (reference example/C/simple.c, ciface:124-234):
 *   - ptr/row are the 1-based CSC pattern of the LOWER triangle;
 *   - akeep/fkeep handles are allocated by spllt_analyse / spllt_all when the
 *     incoming pointer is NULL and released by spllt_deallocate_*;
 *   - user arrays are borrowed; `val` must stay alive until spllt_wait();
 *   - spllt_factor only submits work (HIP streams); spllt_wait() is the
 *     completion barrier and makes L visible to the host-side solve;
 *   - errors never abort: info->flag < 0 (0 ok, -1 allocation, -10 bad
 *     parameter, -98 unimplemented, -99 unknown; reference
 *     src/spllt_data_mod.F90:31-35).  Additive to the reference: -20 = matrix
 *     not positive definite (the reference swallows this,
 *     src/spllt_kernels_mod.F90:1179-1181), -30 = HIP runtime failure.
 */
#ifndef SPLLT_IFACE_H
#define SPLLT_IFACE_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference include/spllt_iface.h:8-12 */
typedef struct {
  void *akeep;
  void *fkeep;
  void *tm;
} spllt_data_t;

/* reference include/spllt_iface.h:14-31 (field order is ABI) */
typedef struct {
  int print_level;
  int nrhs;
  int ncpu;            /* pruning target: number of workers; here number of GPUs */
  int nb;              /* tile size */
  int nemin;           /* supernode amalgamation threshold */
  int prune_tree;      /* mark large independent subtrees (multi-GPU partition) */
  int min_width_blas;  /* accepted for compatibility; the HIP path has one code path */
  int nb_min;
  int nb_max;
  int nrhs_min;
  int nrhs_max;
  int nb_linear_comp;
  int nrhs_linear_comp;
  int chunk;
} spllt_options_t;

/* reference include/spllt_iface.h:33-47 */
#define SPLLT_OPTIONS_NULL()                                                     \
  {                                                                              \
    .print_level = 0, .nrhs = 1, .ncpu = 1, .nb = 16, .nemin = 32,               \
    .prune_tree = 1, .min_width_blas = 8, .nb_min = 32, .nb_max = 32,            \
    .nrhs_min = 1, .nrhs_max = 1, .nb_linear_comp = 0, .nrhs_linear_comp = 0,    \
    .chunk = 10                                                                  \
  }

/* reference include/spllt_iface.h:49-57.  num_factor/num_flops are C ints in
 * the reference ABI and overflow on large problems (ciface:77-78); the exact
 * 64-bit values are available through spllt_hip_sym_info (spllt_hip.h). */
typedef struct {
  int flag;
  int maxdepth;
  int num_factor;
  int num_flops;
  int num_nodes;
  int stat;
} spllt_inform_t;

#define SPLLT_SUCCESS 0
#define SPLLT_ERROR_ALLOCATION (-1)
#define SPLLT_ERROR_PARAMETER (-10)
#define SPLLT_ERROR_NOT_POSDEF (-20)
#define SPLLT_ERROR_HIP (-30)
#define SPLLT_ERROR_UNIMPLEMENTED (-98)
#define SPLLT_ERROR_UNKNOWN (-99)

/* :59  (ciface:124-186)  symbolic analysis; order[i] (1-based) out */
void spllt_analyse(void **akeep, void **fkeep, spllt_options_t *options, int n, int *ptr,
                   int *row, spllt_inform_t *info, int *order);

/* :68  (ciface:190-234)  asynchronous numerical factorization P A P^T = L L^T.
 * `val` must stay valid until spllt_wait returns (the reference reads it from its tasks, too).
 * If the call comes back with flag -30 because the submission itself ran into its deadline (the
 * HIP runtime did not return), `val` must stay valid for the rest of the process: the library's
 * helper thread may still be reading it; the handle is dead and spllt_deallocate_fkeep leaks it. */
void spllt_factor(void *akeep, void *fkeep, spllt_options_t *options, int nnz, double *val,
                  spllt_inform_t *info);

/* :75  (ciface:238-287) */
void spllt_prepare_solve(void *akeep, void *fkeep, int nb, int nrhs, long *worksize,
                         spllt_inform_t *info);

/* :82  (ciface:291-339) */
void spllt_set_mem_solve(void *akeep, void *fkeep, int nb, int nrhs, long worksize, double *y,
                         double *workspace, spllt_inform_t *info);

/* :91  declared by the reference without a definition (ciface:343-369 is
 * commented out); provided here. */
void spllt_solve_workspace_size(void *fkeep, int nworker, int nrhs, long *size);

/* :96  (ciface:372-428)  job: 0 = both, 1 = forward, 2 = backward; `order` is
 * accepted and ignored exactly as in the reference (ciface:404-419). */
void spllt_solve(void *fkeep, spllt_options_t *options, int *order, int nrhs, double *x,
                 spllt_inform_t *info, int job);

/* :104 (ciface:432-498) */
void spllt_solve_worker(void *fkeep, spllt_options_t *options, int *order, int nrhs, double *x,
                        spllt_inform_t *info, int job, double *workspace, long worksize,
                        void *tm);

/* :115 (src/spllt_mod.F90:172-182) completion barrier: drains every stream.  Like the
 * reference's (a bare `!$omp taskwait`) it takes no handle and returns nothing: a failure that
 * only shows when the work has run -- a matrix that is not positive definite (flag -20; the
 * reference swallows dpotrf's info, kernels_mod:1179-1181), a device that does not drain (-30) --
 * is reported by the NEXT call on that fkeep that takes `info` (spllt_solve, spllt_factor, ...),
 * with a message on stderr at once; spllt_hip_last_flag(fkeep) (spllt_hip.h) reads it directly. */
void spllt_wait(void);

/* :117 (ciface:502-552) prints ||Ax-b||/||b|| style checks to stdout */
void spllt_chkerr(int n, int *ptr, int *row, double *val, int nrhs, double *x, double *rhs);

/* :125, :128 (ciface:555-617) */
void spllt_deallocate_fkeep(void **fkeep, int *stat);
void spllt_deallocate_akeep(void **akeep, int *stat);

/* :131, :134 (ciface:621-652, :89-109) */
void spllt_task_manager_deallocate(void **task_manager, int *stat);
void spllt_task_manager_init(void **task_manager);

/* :136 (ciface:656-780) analyse + factor + solve + check in one call */
void spllt_all(void **akeep, void **fkeep, spllt_options_t *options, int n, int nnz, int nrhs,
               int nb, int *ptr, int *row, double *val, double *x, double *rhs,
               spllt_inform_t *info);

#ifdef __cplusplus
}
#endif
#endif
