/*
 * spllt_oracle -- CPU restatement of SpLLT's factorize path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under spllt_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / reported baseline.
 *
 * Parity status: PARITY UNPINNED.  The reference holds no golden output for this path
 * (example/C/simple.c has inputs only; tests/golden/kat_simple_c.json is the closed-form
 * answer of that 3x3 case, typed here) and its Fortran cannot be built in this image: every
 * module of its factor path uses SPRAL's `spral_ssids_inform` module
 * (src/spllt_data_mod.F90:13), SPRAL is absent, and writing a stand-in for it is not
 * permitted -> no oracle/_ref.  What checks this restatement instead: (i) that 3x3 case,
 * (ii) the reference's residual bar ||r||/(||b|| + max|a| ||x||) <= 1e-14
 * (src/utils_mod.F90:462-467) and (iii) agreement with an independent dense LAPACK Cholesky
 * of P A P^T (uniqueness of the Cholesky factor).
 *
 * Each function cites the reference routine (file:line under /root/reference)
 * whose behaviour it restates.  All indices are 0-based here.
 */
#ifndef SPLLT_ORACLE_H
#define SPLLT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* tile descriptor: type spllt_block, src/spllt_data_mod.F90:121-172 */
typedef struct {
  int64_t id, dblk, last_blk; /* tile ids (0-based) */
  int64_t sa;                 /* offset of the tile inside its lcol */
  int bcol, blkm, blkn, node;
} spo_block;

typedef struct spo_factor spo_factor;

/* Build nodes / tiles / lmap from the SSIDS-style symbolic quintuple
 * (restates src/spllt_analyse_mod.F90:305-469 and :1033-1171).
 * order[var] = pivot position.  small may be NULL (no pruned subtrees);
 * otherwise small[node] = 0 | 1 | -(root+1). */
spo_factor *spo_create(int n, int nnodes, const int *sptr, const int *sparent,
                       const int64_t *rptr, const int *rlist, const int *order,
                       const int64_t *ptr, const int *row, int nb,
                       const int *small, int min_width_blas);
void spo_destroy(spo_factor *f);

/* spllt_stf_factorize + spllt_wait (src/spllt_stf_mod.F90:18-192).
 * nthreads <= 1: plain sequential execution in submission order;
 * nthreads  > 1: OpenMP tasks with the reference's dependency tokens.
 * Returns 0, or k>0 if a diagonal tile was not positive definite (first such
 * tile id + 1; the reference ignores this, src/spllt_kernels_mod.F90:1179-1181). */
int spo_factorize(spo_factor *f, const double *val, int nthreads);

/* forward+backward solve with the tiles (x overwritten), restating the math of
 * src/spllt_solve_mod.F90:244-411 without its task machinery; rhs in original
 * variable order. */
void spo_solve(const spo_factor *f, int nrhs, double *x);

int spo_nbcol(const spo_factor *f);
int64_t spo_nblk(const spo_factor *f);
int spo_maxmn(const spo_factor *f);
int64_t spo_arena(const spo_factor *f);          /* sum of lcol sizes */
int64_t spo_lcol_size(const spo_factor *f, int bcol);
const double *spo_lcol(const spo_factor *f, int bcol);
const spo_block *spo_blocks(const spo_factor *f);
int64_t spo_lmap_len(const spo_factor *f, int bcol);
const int64_t *spo_lmap_dst(const spo_factor *f, int bcol); /* offset in lcol */
const int64_t *spo_lmap_src(const spo_factor *f, int bcol); /* index in val  */
/* copy every lcol, concatenated in block-column order, into out[spo_arena] */
void spo_export_arena(const spo_factor *f, double *out);

/* ---- individual kernels (row-major tiles, exactly the reference's calls) ---- */
/* spllt_factor_diag_block, src/spllt_kernels_mod.F90:1168-1189 */
int spo_factor_diag_block(int m, int n, double *dest);
/* spllt_solve_block, :1217-1229 */
void spo_solve_block(int m, int n, double *dest, const double *diag);
/* spllt_update_block, :1261-1292 */
void spo_update_block(int m, int n, double *dest, int diag, int n1,
                      const double *src1, const double *src2);
/* spllt_expand_buffer, :2010-2053 (row_list/col_list 0-based) */
void spo_expand_buffer(double *a, int blkn, const int *row_list, int rls,
                       const int *col_list, int cls, int ndiag, const double *buffer);
/* spllt_update_direct, :14-93 */
void spo_update_direct(int n, double *dest, int n1, const double *csrc, const double *rsrc,
                       const int *row_list, int rls, const int *col_list, int cls, int ndiag);
/* spllt_scatter_block, :1122-1160 */
void spo_scatter_block(int s_m, int s_n, const int *rsrc_index, const int *csrc_index,
                       const double *src, int lds, const int *rdest_index,
                       const int *cdest_index, double *dest, int ldd);
/* spllt_update_between_compute_map, :1606-1723.  Index lists are 0-based
 * pivot positions; outputs 0-based.  Returns 0 if there are no incident
 * columns/rows (lists empty). */
int spo_compute_map(int d_sa, int d_en, int d_nb, const int *d_index, int d_size, int dcol,
                    int dblk_row, /* row-tile index of the dest tile within its block column */
                    int s_sa, int s_en, int s_nb, const int *s_index, int s_size, int scol,
                    int *row_list, int *col_list, int *rls, int *cls,
                    int *s1sa, int *s1en, int *s2sa, int *s2en);

const char *spo_blas_name(void);

#ifdef __cplusplus
}
#endif
#endif
