/* Column-major BLAS subset used by the oracle (see mini_blas.c). TEST INFRASTRUCTURE. */
#ifndef SPO_MINI_BLAS_H
#define SPO_MINI_BLAS_H
#include <stdint.h>
int spo_dpotrf_u(int n, double *a, int lda);
void spo_dtrsm_lutn(int m, int n, const double *a, int lda, double *b, int ldb);
void spo_dsyrk_ut(int n, int k, double alpha, const double *a, int lda, double beta,
                  double *c, int ldc);
void spo_dgemm_tn(int m, int n, int k, double alpha, const double *a, int lda,
                  const double *b, int ldb, double beta, double *c, int ldc);
const char *spo_blas_name(void);
#endif
