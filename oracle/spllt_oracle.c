/*
 * spllt_oracle.c -- CPU restatement of SpLLT's factorize path (see
 * spllt_oracle.h for status and rules: TEST INFRASTRUCTURE ONLY).
 *
 * Layout facts restated from the reference:
 *   - every block column of L is one contiguous array lcol, a row-major
 *     (rows x blkn) matrix; tiles are nb-row slices of it, diagonal first;
 *   - BLAS is called on the column-major transpose, hence 'U','T' / 'T','N'.
 */
#include "spllt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "mini_blas.h"

typedef struct {
  int sa, en; /* pivot positions of own columns, inclusive */
  int nrow;
  const int *index; /* sorted pivot positions, own columns first */
  int nb;
  int64_t blk_sa, blk_en;
  int parent; /* nnodes = virtual root */
  int least_desc;
  double *buffer; /* generated element of a pruned-subtree root */
} node_t;

struct spo_factor {
  int n, nnodes, nb, nbcol, maxmn, min_width_blas;
  int64_t final_blk, arena;
  node_t *nodes;
  spo_block *bc;
  int *small;
  int *rlist;
  int64_t *rptr;
  double **lcol;
  int64_t *lcol_size;
  int64_t *lmap_ptr, *lmap_dst, *lmap_src;
  int *order, *porder;
  int *bcol_dblk; /* diagonal tile of each block column */
  int pd_fail;
};

/* per-thread scratch: spllt_factorization_init, src/spllt_factorization_mod.F90:347-423 */
typedef struct {
  double *workspace; /* maxmn^2 */
  int *row_list, *col_list, *map;
} scratch_t;

static scratch_t *scratch_new(const spo_factor *f) {
  scratch_t *s = (scratch_t *)malloc(sizeof *s);
  size_t mm = (size_t)f->maxmn * f->maxmn;
  s->workspace = (double *)malloc(sizeof(double) * (mm ? mm : 1));
  s->row_list = (int *)malloc(sizeof(int) * (f->maxmn + 1));
  s->col_list = (int *)malloc(sizeof(int) * (f->maxmn + 1));
  s->map = (int *)malloc(sizeof(int) * (f->n + 1));
  return s;
}
static void scratch_free(scratch_t *s) {
  free(s->workspace); free(s->row_list); free(s->col_list); free(s->map); free(s);
}

/* ------------------------------------------------------------------------- */
/* Construction: tiles and val->lcol map                                      */
/* ------------------------------------------------------------------------- */
spo_factor *spo_create(int n, int nnodes, const int *sptr, const int *sparent,
                       const int64_t *rptr, const int *rlist, const int *order,
                       const int64_t *ptr, const int *row, int nb, const int *small,
                       int min_width_blas) {
  spo_factor *f = (spo_factor *)calloc(1, sizeof *f);
  f->n = n; f->nnodes = nnodes; f->nb = nb < 1 ? 256 : nb;
  f->min_width_blas = min_width_blas;
  nb = f->nb;
  f->nodes = (node_t *)calloc((size_t)nnodes + 1, sizeof(node_t));
  f->small = (int *)calloc((size_t)nnodes + 1, sizeof(int));
  if (small) memcpy(f->small, small, sizeof(int) * nnodes);
  f->rptr = (int64_t *)malloc(sizeof(int64_t) * (nnodes + 1));
  memcpy(f->rptr, rptr, sizeof(int64_t) * (nnodes + 1));
  f->rlist = (int *)malloc(sizeof(int) * (rptr[nnodes] ? rptr[nnodes] : 1));
  memcpy(f->rlist, rlist, sizeof(int) * rptr[nnodes]);
  f->order = (int *)malloc(sizeof(int) * (n ? n : 1));
  f->porder = (int *)malloc(sizeof(int) * (n ? n : 1));
  memcpy(f->order, order, sizeof(int) * n);
  for (int i = 0; i < n; ++i) f->porder[order[i]] = i;

  /* node table + tile counts: src/spllt_analyse_mod.F90:305-358 */
  int64_t blk = 0;
  int nbcol = 0;
  for (int s = 0; s < nnodes; ++s) {
    node_t *nd = &f->nodes[s];
    nd->sa = sptr[s]; nd->en = sptr[s + 1] - 1;
    nd->nrow = (int)(rptr[s + 1] - rptr[s]);
    nd->index = f->rlist + rptr[s];
    nd->nb = nb;
    nd->parent = sparent[s];
    nd->least_desc = s;
    int sz = (nd->nrow - 1) / nb + 1, j = 0;
    for (int c = nd->sa; c <= nd->en; c += nb) { j += sz; sz--; nbcol++; }
    nd->blk_sa = blk;
    nd->blk_en = blk + j - 1;
    blk += j;
  }
  for (int s = 0; s < nnodes; ++s) { /* least descendants, :273-290 */
    int p = f->nodes[s].parent;
    if (p < nnodes && f->nodes[s].least_desc < f->nodes[p].least_desc)
      f->nodes[p].least_desc = f->nodes[s].least_desc;
  }
  f->final_blk = blk;
  f->nbcol = nbcol;
  f->bc = (spo_block *)calloc((size_t)(blk ? blk : 1), sizeof(spo_block));
  f->lcol = (double **)calloc((size_t)(nbcol ? nbcol : 1), sizeof(double *));
  f->lcol_size = (int64_t *)calloc((size_t)nbcol + 1, sizeof(int64_t));
  f->bcol_dblk = (int *)calloc((size_t)nbcol + 1, sizeof(int));

  /* tile descriptors: src/spllt_analyse_mod.F90:381-469 */
  blk = 0;
  int bcol = 0;
  for (int s = 0; s < nnodes; ++s) {
    node_t *nd = &f->nodes[s];
    int numcol = nd->en - nd->sa + 1, numrow = nd->nrow;
    int sz = (numrow - 1) / nb + 1, col_used = 0;
    for (int ci = nd->sa; ci <= nd->en; ci += nb) {
      int blkn = numcol - col_used < nb ? numcol - col_used : nb;
      col_used += blkn;
      int64_t dblk = blk, k = 0;
      int row_used = 0;
      f->bcol_dblk[bcol] = (int)dblk;
      for (blk = dblk; blk < dblk + sz; ++blk) {
        spo_block *b = &f->bc[blk];
        b->id = blk;
        b->blkm = numrow - row_used < nb ? numrow - row_used : nb;
        row_used += b->blkm;
        b->blkn = blkn;
        if (b->blkm > f->maxmn) f->maxmn = b->blkm;
        if (b->blkn > f->maxmn) f->maxmn = b->blkn;
        b->sa = k;
        b->dblk = dblk;
        b->last_blk = dblk + sz - 1;
        b->node = s;
        b->bcol = bcol;
        k += (int64_t)b->blkm * b->blkn;
      }
      f->lcol_size[bcol] = k;
      f->arena += k;
      bcol++;
      sz--;
      numrow -= nb;
    }
  }

  /* val -> lcol map: spllt_make_map + spllt_lcol_map, :1033-1171.
   * Entry (i,j) goes to pivot column min(order i, order j). */
  int64_t nz = ptr[n];
  f->lmap_ptr = (int64_t *)calloc((size_t)nbcol + 2, sizeof(int64_t));
  f->lmap_dst = (int64_t *)malloc(sizeof(int64_t) * (nz ? nz : 1));
  f->lmap_src = (int64_t *)malloc(sizeof(int64_t) * (nz ? nz : 1));
  {
    /* reordered lower triangle in CSC by pivot column (nptr/nrow/amap) */
    int64_t *nptr = (int64_t *)calloc((size_t)n + 2, sizeof(int64_t));
    int *nrow = (int *)malloc(sizeof(int) * (nz ? nz : 1));
    int64_t *amap = (int64_t *)malloc(sizeof(int64_t) * (nz ? nz : 1));
    for (int j = 0; j < n; ++j)
      for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) {
        int k = order[row[e]], l = order[j];
        nptr[(k < l ? k : l) + 1]++;
      }
    for (int j = 0; j < n; ++j) nptr[j + 1] += nptr[j];
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    memcpy(pos, nptr, sizeof(int64_t) * (n + 1));
    for (int j = 0; j < n; ++j)
      for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) {
        int k = order[row[e]], l = order[j];
        int c = k < l ? k : l, r = k < l ? l : k;
        amap[pos[c]] = e;
        nrow[pos[c]++] = r;
      }
    int *map = (int *)malloc(sizeof(int) * (n + 1));
    int64_t w = 0;
    for (int s = 0; s < nnodes; ++s) {
      node_t *nd = &f->nodes[s];
      for (int j = 0; j < nd->nrow; ++j) map[nd->index[j]] = j;
      int64_t dblk = nd->blk_sa;
      for (int cb = nd->sa; cb <= nd->en; cb += nb) {
        int bc = f->bc[dblk].bcol;
        int swidth = f->bc[dblk].blkn;
        /* offset so that local row i lands at offset + i*swidth (0-based) */
        int64_t offset = f->bc[dblk].sa - (int64_t)(cb - nd->sa) * swidth;
        int last = cb + nb - 1 < nd->en ? cb + nb - 1 : nd->en;
        f->lmap_ptr[bc] = w;
        for (int col = cb; col <= last; ++col) {
          for (int64_t e = nptr[col]; e < nptr[col + 1]; ++e) {
            f->lmap_dst[w] = offset + (int64_t)map[nrow[e]] * swidth;
            f->lmap_src[w] = amap[e];
            w++;
          }
          offset++;
        }
        dblk = f->bc[dblk].last_blk + 1;
      }
    }
    f->lmap_ptr[nbcol] = w;
    free(map); free(pos); free(amap); free(nrow); free(nptr);
  }
  return f;
}

void spo_destroy(spo_factor *f) {
  if (!f) return;
  for (int b = 0; b < f->nbcol; ++b) free(f->lcol[b]);
  for (int s = 0; s < f->nnodes; ++s) free(f->nodes[s].buffer);
  free(f->lcol); free(f->lcol_size); free(f->bcol_dblk);
  free(f->lmap_ptr); free(f->lmap_dst); free(f->lmap_src);
  free(f->bc); free(f->nodes); free(f->small); free(f->rptr); free(f->rlist);
  free(f->order); free(f->porder);
  free(f);
}

/* ------------------------------------------------------------------------- */
/* Kernels                                                                    */
/* ------------------------------------------------------------------------- */

/* spllt_factor_diag_block: dpotrf('U') of the n x n head, then dtrsm of the
 * m-n trailing rows of a trapezoidal diagonal tile. */
int spo_factor_diag_block(int m, int n, double *dest) {
  int info = spo_dpotrf_u(n, dest, n);
  if (info != 0) return info; /* the reference returns silently here */
  if (m > n) spo_dtrsm_lutn(n, m - n, dest, n, dest + (int64_t)n * n, n);
  return 0;
}

/* spllt_solve_block: dest <- dest * L_kk^-T */
void spo_solve_block(int m, int n, double *dest, const double *diag) {
  spo_dtrsm_lutn(n, m, diag, n, dest, n);
}

/* spllt_update_block: dest -= src2 * src1^T (lower part only on the diagonal) */
void spo_update_block(int m, int n, double *dest, int diag, int n1, const double *src1,
                      const double *src2) {
  if (diag) {
    spo_dsyrk_ut(n, n1, -1.0, src1, n1, 1.0, dest, n);
    if (m > n)
      spo_dgemm_tn(n, m - n, n1, -1.0, src1, n1, src2 + (int64_t)n * n1, n1, 1.0,
                   dest + (int64_t)n * n, n);
  } else {
    spo_dgemm_tn(n, m, n1, -1.0, src1, n1, src2, n1, 1.0, dest, n);
  }
}

/* spllt_expand_buffer: indexed += of a (rls x cls) buffer into a tile; the
 * first ndiag rows only up to their diagonal. */
void spo_expand_buffer(double *a, int blkn, const int *row_list, int rls, const int *col_list,
                       int cls, int ndiag, const double *buffer) {
  for (int j = 0; j < rls; ++j) {
    const double *b = buffer + (int64_t)j * cls;
    double *arow = a + (int64_t)row_list[j] * blkn;
    int imax = j < ndiag ? j + 1 : cls;
    for (int i = 0; i < imax; ++i) arow[col_list[i]] += b[i];
  }
}

/* spllt_update_direct: the same update without a buffer, one dot product of
 * length n1 per destination entry (used when n1 < min_width_blas). */
void spo_update_direct(int n, double *dest, int n1, const double *csrc, const double *rsrc,
                       const int *row_list, int rls, const int *col_list, int cls, int ndiag) {
  for (int j = 0; j < rls; ++j) {
    double *drow = dest + (int64_t)row_list[j] * n;
    const double *r = rsrc + (int64_t)j * n1;
    int imax = j < ndiag ? j + 1 : cls;
    for (int i = 0; i < imax; ++i) {
      const double *c = csrc + (int64_t)i * n1;
      double w = 0.0;
      for (int l = 0; l < n1; ++l) w += c[l] * r[l];
      drow[col_list[i]] -= w;
    }
  }
}

/* spllt_scatter_block: dest -= src, locating every source row/column in the
 * destination's index lists by a forward walk. */
void spo_scatter_block(int s_m, int s_n, const int *rsrc_index, const int *csrc_index,
                       const double *src, int lds, const int *rdest_index,
                       const int *cdest_index, double *dest, int ldd) {
  int dr = 0;
  for (int sr = 0; sr < s_m; ++sr) {
    while (rdest_index[dr] != rsrc_index[sr]) dr++;
    int dc = 0;
    for (int sc = 0; sc < s_n; ++sc) {
      while (cdest_index[dc] != csrc_index[sc]) dc++;
      dest[(int64_t)dr * ldd + dc] -= src[(int64_t)sr * lds + sc];
    }
  }
}

/* spllt_update_between_compute_map: two sorted-list merges. */
int spo_compute_map(int d_sa, int d_en, int d_nb, const int *d_index, int d_size, int dcol,
                    int dblk_row, int s_sa, int s_en, int s_nb, const int *s_index, int s_size,
                    int scol, int *row_list, int *col_list, int *rls, int *cls, int *s1sa,
                    int *s1en, int *s2sa, int *s2en) {
  *rls = 0; *cls = 0;
  int dcsa = d_sa + dcol * d_nb;
  int dcen = d_sa + (dcol + 1) * d_nb - 1;
  if (dcen > d_en) dcen = d_en;
  int sncol = s_en - s_sa + 1;
  int cptr = scol * s_nb < sncol ? scol * s_nb : sncol;
  if (cptr >= s_size) return 0;
  while (s_index[cptr] < dcsa) {
    cptr++;
    if (cptr >= s_size) return 0; /* no incident columns */
  }
  *s1sa = cptr;
  while (s_index[cptr] <= dcen) {
    col_list[(*cls)++] = s_index[cptr] - dcsa;
    cptr++;
    if (cptr >= s_size) break;
  }
  *s1en = cptr - 1;
  int i = dcol + dblk_row; /* row-tile index within dnode */
  int drsa = d_index[i * d_nb];
  int last = (i + 1) * d_nb - 1;
  if (last > d_size - 1) last = d_size - 1;
  int dren = d_index[last];
  int rptr = *s1sa;
  while (s_index[rptr] < drsa) {
    rptr++;
    if (rptr >= s_size) return 0;
  }
  *s2sa = rptr;
  int dptr_sa = i * d_nb, dptr = dptr_sa;
  for (rptr = *s2sa; rptr < s_size; ++rptr) {
    if (s_index[rptr] > dren) break;
    while (d_index[dptr] < s_index[rptr]) dptr++;
    row_list[(*rls)++] = dptr - dptr_sa;
  }
  *s2en = rptr - 1;
  return 1;
}

/* spllt_update_between, src/spllt_kernels_mod.F90:2108-2237 */
static void update_between(const spo_factor *f, const spo_block *blk, int dcol,
                           const node_t *dnode, int n1, int scol, const node_t *snode,
                           double *dest, const double *csrc, const double *rsrc,
                           scratch_t *sc) {
  int rls, cls, s1sa = 0, s1en = -1, s2sa = 0, s2en = -1;
  int n = blk->blkn;
  int diag = blk->dblk == blk->id;
  spo_compute_map(dnode->sa, dnode->en, dnode->nb, dnode->index, dnode->nrow, dcol,
                  (int)(blk->id - blk->dblk), snode->sa, snode->en, snode->nb, snode->index,
                  snode->nrow, scol, sc->row_list, sc->col_list, &rls, &cls, &s1sa, &s1en,
                  &s2sa, &s2en);
  if (rls == 0 || cls == 0) return;
  if (n1 >= f->min_width_blas) {
    double *buffer = sc->workspace;
    int ndiag;
    if (diag) {
      ndiag = s1en - s1sa + 1;
      spo_dsyrk_ut(ndiag, n1, -1.0, csrc, n1, 0.0, buffer, cls);
      if (s2en - s2sa + 1 - ndiag > 0)
        spo_dgemm_tn(ndiag, s2en - s2sa + 1 - ndiag, n1, -1.0, csrc, n1,
                     rsrc + (int64_t)n1 * ndiag, n1, 0.0, buffer + (int64_t)cls * ndiag, cls);
    } else {
      ndiag = 0;
      spo_dgemm_tn(s1en - s1sa + 1, s2en - s2sa + 1, n1, -1.0, csrc, n1, rsrc, n1, 0.0,
                   buffer, cls);
    }
    spo_expand_buffer(dest, n, sc->row_list, rls, sc->col_list, cls, ndiag, buffer);
  } else {
    int ndiag = diag ? s1en - s1sa + 1 : 0;
    spo_update_direct(n, dest, n1, csrc, rsrc, sc->row_list, rls, sc->col_list, cls, ndiag);
  }
}

/* spllt_init_node, src/spllt_kernels_mod.F90:2301-2364: zero every lcol of the
 * node, then lcol(map(1,i)) = val(map(2,i)) (assignment). */
static void init_node(spo_factor *f, int s, const double *val) {
  const node_t *nd = &f->nodes[s];
  int64_t dblk = nd->blk_sa;
  while (dblk <= nd->blk_en) {
    int bc = f->bc[dblk].bcol;
    memset(f->lcol[bc], 0, sizeof(double) * f->lcol_size[bc]);
    for (int64_t i = f->lmap_ptr[bc]; i < f->lmap_ptr[bc + 1]; ++i)
      f->lcol[bc][f->lmap_dst[i]] = val[f->lmap_src[i]];
    dblk = f->bc[dblk].last_blk + 1;
  }
}

static inline double *tile_ptr(const spo_factor *f, int64_t blk) {
  return f->lcol[f->bc[blk].bcol] + f->bc[blk].sa;
}

/* get_dest_block, src/spllt_data_mod.F90:663-683 */
static int64_t get_dest_block(const spo_block *src1, const spo_block *src2) {
  int64_t sz = src1->last_blk - src1->dblk + 1, d = src1->dblk;
  for (int64_t i = src1->dblk + 1; i <= src1->id; ++i) { d += sz; sz--; }
  return d + src2->id - src1->id;
}

/* The row -> row-tile map of an ancestor: spllt_build_rowmap, :2519-2546 */
static void build_rowmap(const node_t *nd, int *map) {
  for (int r = 0; r < nd->nrow; ++r) map[nd->index[r]] = r / nd->nb;
}

/* ---- task bodies: src/spllt_factorization_task_mod.F90 ------------------- */
static void factorize_block_task(spo_factor *f, int64_t dblk) { /* :351 */
  double *d = tile_ptr(f, dblk);
#pragma omp task firstprivate(f, dblk, d) depend(inout : d[0])
  {
    int info = spo_factor_diag_block(f->bc[dblk].blkm, f->bc[dblk].blkn, d);
    if (info) {
#pragma omp critical(spo_pd)
      if (!f->pd_fail) f->pd_fail = (int)dblk + 1;
    }
  }
}
static void solve_block_task(spo_factor *f, int64_t dblk, int64_t blk) { /* :482 */
  double *d = tile_ptr(f, dblk), *x = tile_ptr(f, blk);
#pragma omp task firstprivate(f, dblk, blk, d, x) depend(in : d[0]) depend(inout : x[0])
  spo_solve_block(f->bc[blk].blkm, f->bc[blk].blkn, x, d);
}
static void update_block_task(spo_factor *f, int64_t ik, int64_t jk, int64_t ij) { /* :648 */
  double *a = tile_ptr(f, ik), *b = tile_ptr(f, jk), *c = tile_ptr(f, ij);
#pragma omp task firstprivate(f, ik, jk, ij, a, b, c) depend(in : a[0], b[0]) depend(inout : c[0])
  spo_update_block(f->bc[ij].blkm, f->bc[ij].blkn, c, f->bc[ij].dblk == f->bc[ij].id,
                   f->bc[jk].blkn, b, a);
}

static scratch_t **g_scratch; /* one per OpenMP thread */
static scratch_t *my_scratch(void) {
#ifdef _OPENMP
  return g_scratch[omp_get_thread_num()];
#else
  return g_scratch[0];
#endif
}

/* spllt_update_between_task, :892-1321.  Source slices: rows cptr..cptr2
 * (-> dest columns) and rptr..rptr2 (-> dest rows) of source block column
 * scol, i.e. offsets (row - scol*s_nb)*n1 into its lcol (:1215-1219). */
static void update_between_task(spo_factor *f, int64_t bc_kk, const node_t *snode,
                                int64_t a_blk, const node_t *anode, int cptr, int cptr2,
                                int rptr, int rptr2) {
  const spo_block *src = &f->bc[bc_kk];
  const spo_block *dst = &f->bc[a_blk];
  int n1 = src->blkn;
  int scol = src->bcol - f->bc[snode->blk_sa].bcol;
  int dcol = dst->bcol - f->bc[anode->blk_sa].bcol;
  double *lcol1 = f->lcol[src->bcol];
  const double *csrc = lcol1 + (int64_t)(cptr - scol * snode->nb) * n1;
  const double *rsrc = lcol1 + (int64_t)(rptr - scol * snode->nb) * n1;
  double *dest = tile_ptr(f, a_blk);
  /* dependency tokens: first and last tile of each source range, :1239-1241 */
  int64_t t_c1 = src->dblk + cptr / snode->nb - scol, t_c2 = src->dblk + cptr2 / snode->nb - scol;
  int64_t t_r1 = src->dblk + rptr / snode->nb - scol, t_r2 = src->dblk + rptr2 / snode->nb - scol;
  double *pc1 = tile_ptr(f, t_c1), *pc2 = tile_ptr(f, t_c2);
  double *pr1 = tile_ptr(f, t_r1), *pr2 = tile_ptr(f, t_r2);
  (void)pc1; (void)pc2; (void)pr1; (void)pr2; (void)cptr2; (void)rptr2;
#pragma omp task firstprivate(f, dst, dcol, anode, n1, scol, snode, dest, csrc, rsrc) \
    depend(in : pc1[0], pc2[0], pr1[0], pr2[0]) depend(inout : dest[0])
  update_between(f, dst, dcol, anode, n1, scol, snode, dest, csrc, rsrc, my_scratch());
}

/* spllt_factorize_node, src/spllt_factorization_mod.F90:474-563 */
static void factorize_node(spo_factor *f, const node_t *nd) {
  int numcol = nd->en - nd->sa + 1, numrow = nd->nrow, nb = nd->nb;
  int nc = (numcol - 1) / nb + 1, nr = (numrow - 1) / nb + 1;
  int64_t dblk = nd->blk_sa;
  for (int kk = 0; kk < nc; ++kk) {
    factorize_block_task(f, dblk);
    for (int ii = kk + 1; ii < nr; ++ii) solve_block_task(f, dblk, dblk + ii - kk);
    for (int jj = kk + 1; jj < nc; ++jj) {
      int64_t blk2 = dblk + jj - kk;
      for (int ii = jj; ii < nr; ++ii) {
        int64_t blk1 = dblk + ii - kk;
        int64_t ij = get_dest_block(&f->bc[blk2], &f->bc[blk1]);
        update_block_task(f, blk1, blk2, ij);
      }
    }
    dblk = f->bc[dblk].last_blk + 1;
  }
}

/* Walk the ancestors of `nd` (up to and including node `stop`), and for each
 * (ancestor tile, source block column) call `emit`.  Shared by
 * spllt_factorize_apply_node (factorization_mod:567-751) and the in-subtree
 * half of spllt_subtree_apply_node (kernels_mod:328-558).  Returns the final
 * cptr (first row of nd that maps above `stop`). */
static int apply_walk(spo_factor *f, const node_t *nd, int stop, int *map) {
  int numcol = nd->en - nd->sa + 1, numrow = nd->nrow, nb = nd->nb;
  int nc = (numcol - 1) / nb + 1;
  int a_num = nd->parent;
  int cptr = numcol;
  while (a_num < f->nnodes && a_num <= stop) {
    const node_t *an = &f->nodes[a_num];
    while (cptr < numrow && nd->index[cptr] < an->sa) cptr++;
    if (cptr >= numrow) break;
    int map_done = 0;
    for (;;) { /* block columns of anode touched by nd */
      if (cptr >= numrow) break;
      if (nd->index[cptr] > an->en) break;
      int cb = (nd->index[cptr] - an->sa) / an->nb;
      int64_t a_dblk = an->blk_sa;
      for (int jb = 0; jb < cb; ++jb) a_dblk = f->bc[a_dblk].last_blk + 1;
      int jlast = an->sa + (cb + 1) * an->nb - 1;
      if (jlast > an->en) jlast = an->en;
      int cptr2 = cptr;
      while (cptr2 < numrow && nd->index[cptr2] <= jlast) cptr2++;
      cptr2--;
      if (!map_done) { build_rowmap(an, map); map_done = 1; }
      int ii = map[nd->index[cptr]], ilast = cptr, i;
      for (i = cptr; i < numrow; ++i) {
        int k = map[nd->index[i]];
        if (k != ii) {
          int64_t a_blk = a_dblk + ii - cb;
          int64_t dblk = nd->blk_sa;
          for (int kk = 0; kk < nc; ++kk) {
            update_between_task(f, dblk, nd, a_blk, an, cptr, cptr2, ilast, i - 1);
            dblk = f->bc[dblk].last_blk + 1;
          }
          ii = k;
          ilast = i;
        }
      }
      {
        int64_t a_blk = a_dblk + ii - cb;
        int64_t dblk = nd->blk_sa;
        for (int kk = 0; kk < nc; ++kk) {
          update_between_task(f, dblk, nd, a_blk, an, cptr, cptr2, ilast, i - 1);
          dblk = f->bc[dblk].last_blk + 1;
        }
      }
      cptr = cptr2 + 1;
    }
    a_num = an->parent;
  }
  return cptr;
}

/* Rows of `nd` that map above the subtree root: accumulate +L_r L_c^T over all
 * source block columns into `workspace`, then add it into the root's generated
 * element (second half of spllt_subtree_apply_node, kernels_mod:565-776, and
 * spllt_subtree_expand_buffer, :225-325).  The diagonal test uses the current
 * row group (the reference tests the following group's index, SURVEY.md
 * Appendix B; the lower triangle receives identical sums either way). */
static void apply_to_buffer(spo_factor *f, const node_t *nd, const node_t *root, int cptr,
                            scratch_t *sc, double *buffer) {
  int numcol = nd->en - nd->sa + 1, numrow = nd->nrow, nb = nd->nb;
  int nc = (numcol - 1) / nb + 1;
  int am = root->nrow, an = root->en - root->sa + 1, b_sz = am - an;
  int *map = sc->map, *col_list = sc->col_list, *row_list = sc->row_list;
  double *W = sc->workspace;
  int buff_col = 0, map_done = 0;
  while (cptr < numrow) {
    while (root->index[buff_col] != nd->index[cptr]) buff_col++;
    int cb = (buff_col - an) / root->nb;
    int jl = (cb + 1) * root->nb < b_sz ? (cb + 1) * root->nb : b_sz;
    int jlast = an + jl - 1;
    int cptr2 = cptr;
    while (cptr2 < numrow && nd->index[cptr2] <= root->index[jlast]) cptr2++;
    cptr2--;
    int acol = buff_col;
    for (int j = cptr; j <= cptr2; ++j) {
      while (root->index[acol] != nd->index[j]) acol++;
      col_list[j - cptr] = acol;
    }
    int m = cptr2 - cptr + 1;
    if (!map_done) {
      for (int r = an; r < am; ++r) map[root->index[r]] = (r - an) / root->nb;
      map_done = 1;
    }
    int i = cptr;
    while (i < numrow) {
      int ilast = i, grp = map[nd->index[i]];
      while (i < numrow && map[nd->index[i]] == grp) i++;
      int n = i - ilast;
      int is_diag = (grp == cb);
      int64_t dblk = nd->blk_sa;
      for (int kk = 0; kk < nc; ++kk) {
        int n1 = f->bc[dblk].blkn;
        const double *lc = f->lcol[f->bc[dblk].bcol];
        const double *csrc = lc + (int64_t)(cptr - kk * nb) * n1;
        const double *rsrc = lc + (int64_t)(ilast - kk * nb) * n1;
        double beta = kk == 0 ? 0.0 : 1.0;
        if (is_diag) {
          spo_dsyrk_ut(m, n1, 1.0, csrc, n1, beta, W, m);
          if (n - m > 0)
            spo_dgemm_tn(m, n - m, n1, 1.0, csrc, n1, rsrc + (int64_t)n1 * m, n1, beta,
                         W + (int64_t)m * m, m);
        } else {
          spo_dgemm_tn(m, n, n1, 1.0, csrc, n1, rsrc, n1, beta, W, m);
        }
        dblk = f->bc[dblk].last_blk + 1;
      }
      int arow = 0, ndiag = is_diag ? m : 0;
      for (int r = 0; r < n; ++r) {
        while (root->index[arow] != nd->index[ilast + r]) arow++;
        row_list[r] = arow;
      }
      for (int r = 0; r < n; ++r) {
        double *brow = buffer + (int64_t)(row_list[r] - an) * b_sz;
        int imax = r < ndiag ? r + 1 : m;
        for (int j = 0; j < imax; ++j) brow[col_list[j] - an] += W[(int64_t)r * m + j];
      }
    }
    cptr = cptr2 + 1;
  }
}

/* spllt_subtree_apply_buffer, src/spllt_factorization_mod.F90:39-191: scatter
 * the generated element of a subtree root into its ancestors' tiles. */
static void subtree_apply_buffer(spo_factor *f, int root, int *map) {
  const node_t *rt = &f->nodes[root];
  int m = rt->nrow, n = rt->en - rt->sa + 1, lds = m - n;
  if (lds == 0) return;
  const double *buffer = rt->buffer;
  int anode = rt->parent, cptr = n;
  while (anode < f->nnodes) {
    const node_t *an = &f->nodes[anode];
    while (cptr < m && rt->index[cptr] < an->sa) cptr++;
    if (cptr >= m) break;
    int map_done = 0, a_nb = an->nb;
    for (;;) {
      if (cptr >= m) break;
      if (rt->index[cptr] > an->en) break;
      int cb = (rt->index[cptr] - an->sa) / a_nb;
      int64_t dblk = an->blk_sa;
      for (int jb = 0; jb < cb; ++jb) dblk = f->bc[dblk].last_blk + 1;
      int jlast = an->sa + (cb + 1) * a_nb - 1;
      if (jlast > an->en) jlast = an->en;
      int cptr2 = cptr;
      while (cptr2 < m && rt->index[cptr2] <= jlast) cptr2++;
      cptr2--;
      if (!map_done) { build_rowmap(an, map); map_done = 1; }
      int i = cptr;
      while (i < m) {
        int ilast = i, jb = map[rt->index[i]];
        while (i < m && map[rt->index[i]] == jb) i++;
        int64_t dest = dblk + jb - cb;
        double *d = tile_ptr(f, dest);
        const double *src = buffer + (int64_t)(ilast - n) * lds + (cptr - n);
        /* spllt_scatter_block_task, src/spllt_factorization_task_mod.F90:14-114 */
#pragma omp task firstprivate(f, rt, an, ilast, i, cptr, cptr2, jb, cb, a_nb, src, lds, d, dest) \
    depend(in : buffer[0]) depend(inout : d[0])
        spo_scatter_block(i - ilast, cptr2 - cptr + 1, rt->index + ilast, rt->index + cptr, src,
                          lds, an->index + jb * a_nb, an->index + cb * a_nb, d,
                          f->bc[dest].blkn);
      }
      cptr = cptr2 + 1;
    }
    anode = an->parent;
  }
}

/* spllt_subtree_factorize, src/spllt_kernels_mod.F90:780-821: ONE task that
 * initialises, factorizes (spllt_subtree_factorize_node, :97-222 = the node
 * DAG without tasks) and right-looks every node of a pruned subtree. */
static void subtree_factorize(spo_factor *f, int root, const double *val, scratch_t *sc) {
  node_t *rt = &f->nodes[root];
  int m = rt->nrow, n = rt->en - rt->sa + 1;
  memset(rt->buffer, 0, sizeof(double) * (size_t)(m - n) * (m - n));
  for (int s = rt->least_desc; s <= root; ++s) init_node(f, s, val);
  for (int s = rt->least_desc; s <= root; ++s) {
    const node_t *nd = &f->nodes[s];
    /* inside a task no further tasks are created: run bodies inline */
    int numcol = nd->en - nd->sa + 1, nb = nd->nb;
    int nc = (numcol - 1) / nb + 1, nr = (nd->nrow - 1) / nb + 1;
    int64_t dblk = nd->blk_sa;
    for (int kk = 0; kk < nc; ++kk) {
      int info = spo_factor_diag_block(f->bc[dblk].blkm, f->bc[dblk].blkn, tile_ptr(f, dblk));
      if (info) {
#pragma omp critical(spo_pd)
        if (!f->pd_fail) f->pd_fail = (int)dblk + 1;
      }
      for (int ii = kk + 1; ii < nr; ++ii)
        spo_solve_block(f->bc[dblk + ii - kk].blkm, f->bc[dblk].blkn,
                        tile_ptr(f, dblk + ii - kk), tile_ptr(f, dblk));
      for (int jj = kk + 1; jj < nc; ++jj)
        for (int ii = jj; ii < nr; ++ii) {
          int64_t ik = dblk + ii - kk, jk = dblk + jj - kk;
          int64_t ij = get_dest_block(&f->bc[jk], &f->bc[ik]);
          spo_update_block(f->bc[ij].blkm, f->bc[ij].blkn, tile_ptr(f, ij),
                           f->bc[ij].dblk == f->bc[ij].id, f->bc[jk].blkn, tile_ptr(f, jk),
                           tile_ptr(f, ik));
        }
      dblk = f->bc[dblk].last_blk + 1;
    }
    /* right-looking updates inside the subtree, then into the generated element */
    {
      int numrow = nd->nrow;
      int a_num = nd->parent, cptr = numcol;
      int *map = sc->map;
      while (a_num < f->nnodes && a_num <= root) {
        const node_t *an = &f->nodes[a_num];
        while (cptr < numrow && nd->index[cptr] < an->sa) cptr++;
        if (cptr >= numrow) break;
        int map_done = 0;
        for (;;) {
          if (cptr >= numrow) break;
          if (nd->index[cptr] > an->en) break;
          int cb = (nd->index[cptr] - an->sa) / an->nb;
          int64_t a_dblk = an->blk_sa;
          for (int jb = 0; jb < cb; ++jb) a_dblk = f->bc[a_dblk].last_blk + 1;
          int jlast = an->sa + (cb + 1) * an->nb - 1;
          if (jlast > an->en) jlast = an->en;
          int cptr2 = cptr;
          while (cptr2 < numrow && nd->index[cptr2] <= jlast) cptr2++;
          cptr2--;
          if (!map_done) { build_rowmap(an, map); map_done = 1; }
          int i = cptr;
          while (i < numrow) {
            int ilast = i, ii = map[nd->index[i]];
            while (i < numrow && map[nd->index[i]] == ii) i++;
            int64_t a_blk = a_dblk + ii - cb;
            int64_t sblk = nd->blk_sa;
            for (int kk = 0; kk < nc; ++kk) {
              const spo_block *src = &f->bc[sblk];
              int n1 = src->blkn;
              const double *lc = f->lcol[src->bcol];
              update_between(f, &f->bc[a_blk], cb, an, n1, kk, nd, tile_ptr(f, a_blk),
                             lc + (int64_t)(cptr - kk * nb) * n1,
                             lc + (int64_t)(ilast - kk * nb) * n1, sc);
              sblk = src->last_blk + 1;
            }
          }
          cptr = cptr2 + 1;
        }
        a_num = an->parent;
      }
      apply_to_buffer(f, nd, rt, cptr, sc, rt->buffer);
    }
  }
}

int spo_factorize(spo_factor *f, const double *val, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  f->pd_fail = 0;
  /* spllt_factorization_init + spllt_activate_node: allocate lfact/lcol and
   * per-thread scratch (factorization_mod:347-423, kernels_mod:2446-2516) */
  for (int b = 0; b < f->nbcol; ++b)
    if (!f->lcol[b]) f->lcol[b] = (double *)malloc(sizeof(double) * (f->lcol_size[b] ? f->lcol_size[b] : 1));
  for (int s = 0; s < f->nnodes; ++s)
    if (f->small[s] == 1 && !f->nodes[s].buffer) {
      size_t b = (size_t)(f->nodes[s].nrow - (f->nodes[s].en - f->nodes[s].sa + 1));
      f->nodes[s].buffer = (double *)malloc(sizeof(double) * (b * b ? b * b : 1));
    }
  g_scratch = (scratch_t **)malloc(sizeof(scratch_t *) * nthreads);
  for (int t = 0; t < nthreads; ++t) g_scratch[t] = scratch_new(f);
  int *map = (int *)malloc(sizeof(int) * (f->n + 1));

#pragma omp parallel num_threads(nthreads)
#pragma omp single
  {
    /* init tasks for nodes outside pruned subtrees: stf_mod:113-131 */
    for (int s = 0; s < f->nnodes; ++s)
      if (f->small[s] == 0) {
#pragma omp task firstprivate(f, s, val)
        init_node(f, s, val);
      }
#pragma omp taskwait
    /* postorder loop: stf_mod:134-159 */
    for (int s = 0; s < f->nnodes; ++s) {
      if (f->small[s] < 0) continue;
      const node_t *nd = &f->nodes[s];
      if (f->small[s] == 1) {
        double *buffer = f->nodes[s].buffer;
        (void)buffer;
#pragma omp task firstprivate(f, s, val) depend(out : buffer[0])
        subtree_factorize(f, s, val, my_scratch());
        subtree_apply_buffer(f, s, map);
      } else {
        factorize_node(f, nd);
        apply_walk(f, nd, f->nnodes, map);
      }
    }
#pragma omp taskwait
  }
  free(map);
  for (int t = 0; t < nthreads; ++t) scratch_free(g_scratch[t]);
  free(g_scratch);
  g_scratch = NULL;
  return f->pd_fail;
}

/* ------------------------------------------------------------------------- */
/* Solve: L y = P b, L^T z = y, x = P^T z (tiles read exactly as stored)       */
/* ------------------------------------------------------------------------- */
void spo_solve(const spo_factor *f, int nrhs, double *x) {
  int n = f->n;
  double *y = (double *)malloc(sizeof(double) * (n ? n : 1));
  for (int r = 0; r < nrhs; ++r) {
    double *xr = x + (int64_t)r * n;
    for (int i = 0; i < n; ++i) y[f->order[i]] = xr[i];
    for (int s = 0; s < f->nnodes; ++s) { /* forward */
      const node_t *nd = &f->nodes[s];
      int nb = nd->nb, c = 0;
      for (int64_t dblk = nd->blk_sa; dblk <= nd->blk_en; dblk = f->bc[dblk].last_blk + 1, ++c) {
        const double *lc = f->lcol[f->bc[dblk].bcol];
        int w = f->bc[dblk].blkn, r0 = c * nb, rows = nd->nrow - r0;
        for (int j = 0; j < w; ++j) {
          double v = y[nd->index[r0 + j]] / lc[(int64_t)j * w + j];
          y[nd->index[r0 + j]] = v;
          for (int i = j + 1; i < rows; ++i) y[nd->index[r0 + i]] -= lc[(int64_t)i * w + j] * v;
        }
      }
    }
    for (int s = f->nnodes - 1; s >= 0; --s) { /* backward */
      const node_t *nd = &f->nodes[s];
      int nb = nd->nb;
      int nc = (nd->en - nd->sa) / nb + 1;
      int64_t *dbl = (int64_t *)malloc(sizeof(int64_t) * nc);
      int c = 0;
      for (int64_t dblk = nd->blk_sa; dblk <= nd->blk_en; dblk = f->bc[dblk].last_blk + 1) dbl[c++] = dblk;
      for (c = nc - 1; c >= 0; --c) {
        const double *lc = f->lcol[f->bc[dbl[c]].bcol];
        int w = f->bc[dbl[c]].blkn, r0 = c * nb, rows = nd->nrow - r0;
        for (int j = w - 1; j >= 0; --j) {
          double v = y[nd->index[r0 + j]];
          for (int i = j + 1; i < rows; ++i) v -= lc[(int64_t)i * w + j] * y[nd->index[r0 + i]];
          y[nd->index[r0 + j]] = v / lc[(int64_t)j * w + j];
        }
      }
      free(dbl);
    }
    for (int i = 0; i < n; ++i) xr[i] = y[f->order[i]];
  }
  free(y);
}

/* ------------------------------------------------------------------------- */
int spo_nbcol(const spo_factor *f) { return f->nbcol; }
int64_t spo_nblk(const spo_factor *f) { return f->final_blk; }
int spo_maxmn(const spo_factor *f) { return f->maxmn; }
int64_t spo_arena(const spo_factor *f) { return f->arena; }
int64_t spo_lcol_size(const spo_factor *f, int b) { return f->lcol_size[b]; }
const double *spo_lcol(const spo_factor *f, int b) { return f->lcol[b]; }
const spo_block *spo_blocks(const spo_factor *f) { return f->bc; }
int64_t spo_lmap_len(const spo_factor *f, int b) { return f->lmap_ptr[b + 1] - f->lmap_ptr[b]; }
const int64_t *spo_lmap_dst(const spo_factor *f, int b) { return f->lmap_dst + f->lmap_ptr[b]; }
const int64_t *spo_lmap_src(const spo_factor *f, int b) { return f->lmap_src + f->lmap_ptr[b]; }
void spo_export_arena(const spo_factor *f, double *out) {
  int64_t o = 0;
  for (int b = 0; b < f->nbcol; ++b) {
    if (f->lcol[b]) memcpy(out + o, f->lcol[b], sizeof(double) * f->lcol_size[b]);
    o += f->lcol_size[b];
  }
}
