/*
 * The four BLAS/LAPACK entry points SpLLT's factor kernels call, restricted to
 * the exact argument combinations they use (src/spllt_kernels_mod.F90:1179,
 * 1185, 1226, 1280-1289, 2200-2212):
 *     dpotrf('U')   dtrsm('L','U','T','N')   dsyrk('U','T')   dgemm('T','N')
 * TEST INFRASTRUCTURE (see spllt_oracle.h).  Built either as plain C loops
 * (default; published textbook algorithms, column-major like the Fortran
 * originals) or, with -DSPO_USE_MKL, forwarded to the vendor library that the
 * reference would link (third-party and unpinned there: CMakeLists.txt:701-753).
 */
#include "mini_blas.h"

#include <math.h>

#ifdef SPO_USE_MKL
extern void dpotrf_(const char *, const int *, double *, const int *, int *);
extern void dtrsm_(const char *, const char *, const char *, const char *, const int *,
                   const int *, const double *, const double *, const int *, double *,
                   const int *);
extern void dsyrk_(const char *, const char *, const int *, const int *, const double *,
                   const double *, const int *, const double *, double *, const int *);
extern void dgemm_(const char *, const char *, const int *, const int *, const int *,
                   const double *, const double *, const int *, const double *, const int *,
                   const double *, double *, const int *);

const char *spo_blas_name(void) { return "mkl-sequential"; }

int spo_dpotrf_u(int n, double *a, int lda) {
  int info = 0;
  dpotrf_("U", &n, a, &lda, &info);
  return info;
}
void spo_dtrsm_lutn(int m, int n, const double *a, int lda, double *b, int ldb) {
  const double one = 1.0;
  dtrsm_("L", "U", "T", "N", &m, &n, &one, a, &lda, b, &ldb);
}
void spo_dsyrk_ut(int n, int k, double alpha, const double *a, int lda, double beta,
                  double *c, int ldc) {
  dsyrk_("U", "T", &n, &k, &alpha, a, &lda, &beta, c, &ldc);
}
void spo_dgemm_tn(int m, int n, int k, double alpha, const double *a, int lda,
                  const double *b, int ldb, double beta, double *c, int ldc) {
  dgemm_("T", "N", &m, &n, &k, &alpha, a, &lda, b, &ldb, &beta, c, &ldc);
}

#else /* plain C */

const char *spo_blas_name(void) { return "plain-c"; }

/* A = U^T U, U upper triangular, column-major; returns 0 or the 1-based index
 * of the first non-positive pivot (LAPACK dpotrf semantics). */
int spo_dpotrf_u(int n, double *a, int lda) {
  for (int j = 0; j < n; ++j) {
    double *cj = a + (int64_t)j * lda;
    double d = cj[j];
    for (int l = 0; l < j; ++l) d -= cj[l] * cj[l];
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    cj[j] = d;
    for (int i = j + 1; i < n; ++i) {
      double *ci = a + (int64_t)i * lda;
      double s = ci[j];
      for (int l = 0; l < j; ++l) s -= cj[l] * ci[l];
      ci[j] = s / d;
    }
  }
  return 0;
}

/* B <- U^-T B ; U is m x m upper (column-major), B is m x n */
void spo_dtrsm_lutn(int m, int n, const double *a, int lda, double *b, int ldb) {
  for (int c = 0; c < n; ++c) {
    double *x = b + (int64_t)c * ldb;
    for (int i = 0; i < m; ++i) {
      const double *ui = a + (int64_t)i * lda; /* column i of U = row i of U^T */
      double s = x[i];
      for (int l = 0; l < i; ++l) s -= ui[l] * x[l];
      x[i] = s / ui[i];
    }
  }
}

/* C(upper) <- alpha A^T A + beta C ; A is k x n */
void spo_dsyrk_ut(int n, int k, double alpha, const double *a, int lda, double beta,
                  double *c, int ldc) {
  for (int j = 0; j < n; ++j) {
    const double *aj = a + (int64_t)j * lda;
    for (int i = 0; i <= j; ++i) {
      const double *ai = a + (int64_t)i * lda;
      double s = 0.0;
      for (int l = 0; l < k; ++l) s += ai[l] * aj[l];
      double *cij = c + i + (int64_t)j * ldc;
      *cij = (beta == 0.0 ? 0.0 : beta * *cij) + alpha * s;
    }
  }
}

/* C <- alpha A^T B + beta C ; A is k x m, B is k x n, C is m x n */
void spo_dgemm_tn(int m, int n, int k, double alpha, const double *a, int lda,
                  const double *b, int ldb, double beta, double *c, int ldc) {
  for (int j = 0; j < n; ++j) {
    const double *bj = b + (int64_t)j * ldb;
    for (int i = 0; i < m; ++i) {
      const double *ai = a + (int64_t)i * lda;
      double s = 0.0;
      for (int l = 0; l < k; ++l) s += ai[l] * bj[l];
      double *cij = c + i + (int64_t)j * ldc;
      *cij = (beta == 0.0 ? 0.0 : beta * *cij) + alpha * s;
    }
  }
}
#endif
