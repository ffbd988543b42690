/*
 * A C caller in the calling context the reference documents for its C-ABI: the whole
 * analyse / factor / wait / solve / chkerr sequence runs inside an OpenMP parallel
 * region, issued by the thread that owns the `single` construct while the other
 * threads of the team idle at its barrier (reference example/C/simple.c:52-75; there
 * spllt_factor only submits tasks and spllt_wait is the taskwait).  Here spllt_factor
 * enqueues HIP work and spllt_wait drains it; the library must tolerate being driven
 * from whichever thread won the `single`, with the rest of the team alive.
 *
 * Own code against the public header only (include/spllt_iface.h).  Matrix: 5-point
 * Laplacian on a g x g grid (lower triangle, 1-based CSC), rhs = A * 1, so x = 1.
 * Two factorizations of the same pattern (the second with scaled values) exercise the
 * re-factorization path from a possibly different thread.
 * Exit code 0 iff every check passes.
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <spllt_iface.h>

static int build_laplacian(int g, int **ptr_out, int **row_out, double **val_out) {
  const int n = g * g;
  int *ptr = malloc((n + 1) * sizeof(int));
  int *row = malloc(3 * (size_t)n * sizeof(int));
  double *val = malloc(3 * (size_t)n * sizeof(double));
  int nnz = 0;
  for (int j = 0; j < n; ++j) {          /* column j: diagonal, east neighbour, south neighbour */
    const int x = j % g, y = j / g;
    ptr[j] = nnz + 1;
    row[nnz] = j + 1; val[nnz++] = 4.0;
    if (x + 1 < g) { row[nnz] = j + 2; val[nnz++] = -1.0; }
    if (y + 1 < g) { row[nnz] = j + g + 1; val[nnz++] = -1.0; }
  }
  ptr[n] = nnz + 1;
  *ptr_out = ptr; *row_out = row; *val_out = val;
  return nnz;
}

/* y = A x for the symmetric matrix given by its lower triangle */
static void symv(int n, const int *ptr, const int *row, const double *val, const double *x, double *y) {
  memset(y, 0, n * sizeof(double));
  for (int j = 0; j < n; ++j)
    for (int k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
      const int i = row[k] - 1;
      y[i] += val[k] * x[j];
      if (i != j) y[j] += val[k] * x[i];
    }
}

int main(int argc, char **argv) {
  const int g = argc > 1 ? atoi(argv[1]) : 40;
  const int n = g * g, nrhs = 2, nb = 16;
  int *ptr, *row;
  double *val;
  const int nnz = build_laplacian(g, &ptr, &row, &val);
  int *order = malloc(n * sizeof(int));
  double *ones = malloc(n * sizeof(double));
  double *rhs = malloc((size_t)n * nrhs * sizeof(double));
  double *x = malloc((size_t)n * nrhs * sizeof(double));
  for (int i = 0; i < n; ++i) ones[i] = 1.0;
  symv(n, ptr, row, val, ones, rhs);
  for (int i = 0; i < n; ++i) rhs[n + i] = 2.0 * rhs[i];       /* second right-hand side: x = 2 */
  memcpy(x, rhs, (size_t)n * nrhs * sizeof(double));

  void *akeep = NULL, *fkeep = NULL;
  spllt_inform_t info;
  spllt_options_t options = SPLLT_OPTIONS_NULL();
  options.nb = nb;
  int stat = 0, fail = 0, single_thread = -1, team = 1;
  double err1 = 0.0, err2 = 0.0;

#pragma omp parallel
  {
#pragma omp single
    {
      single_thread = omp_get_thread_num();
      team = omp_get_num_threads();
      spllt_analyse(&akeep, &fkeep, &options, n, ptr, row, &info, order);
      if (info.flag < 0) fail |= 1;
      spllt_factor(akeep, fkeep, &options, nnz, val, &info);
      if (info.flag < 0) fail |= 2;
      spllt_wait();

      long worksize = 0;
      spllt_prepare_solve(akeep, fkeep, nb, nrhs, &worksize, &info);
      double *y = calloc((size_t)n * nrhs, sizeof(double));
      double *workspace = calloc(worksize > 0 ? (size_t)worksize : 1, sizeof(double));
      spllt_set_mem_solve(akeep, fkeep, nb, nrhs, worksize, y, workspace, &info);
      spllt_solve(fkeep, &options, order, nrhs, x, &info, 0);
      if (info.flag < 0) fail |= 4;
      spllt_wait();
      spllt_chkerr(n, ptr, row, val, nrhs, x, rhs);
      for (int i = 0; i < n; ++i) {
        err1 = fmax(err1, fabs(x[i] - 1.0));
        err1 = fmax(err1, fabs(x[n + i] - 2.0));
      }
      free(y);
      free(workspace);
    }
    /* second factorization of the same pattern, submitted by the LAST thread of the team */
#pragma omp barrier
    if (omp_get_thread_num() == omp_get_num_threads() - 1) {
      double *val4 = malloc(nnz * sizeof(double));
      for (int k = 0; k < nnz; ++k) val4[k] = 4.0 * val[k];
      spllt_factor(akeep, fkeep, &options, nnz, val4, &info);
      if (info.flag < 0) fail |= 8;
      spllt_wait();
      memcpy(x, rhs, (size_t)n * sizeof(double));
      spllt_solve(fkeep, &options, order, 1, x, &info, 0);   /* (4A) x = A 1  ->  x = 1/4 */
      if (info.flag < 0) fail |= 16;
      for (int i = 0; i < n; ++i) err2 = fmax(err2, fabs(x[i] - 0.25));
      free(val4);
    }
  }
  spllt_deallocate_akeep(&akeep, &stat);
  if (stat != 0 || akeep != NULL) fail |= 32;
  spllt_deallocate_fkeep(&fkeep, &stat);
  if (stat != 0 || fkeep != NULL) fail |= 64;
  if (!(err1 <= 1e-10)) fail |= 128;
  if (!(err2 <= 1e-10)) fail |= 256;
  printf("omp_caller: n=%d nnz=%d team=%d single_on_thread=%d max|x-1|=%.2e max|x-1/4|=%.2e fail=%d\n", n, nnz,
         team, single_thread, err1, err2, fail);
  free(ptr); free(row); free(val); free(order); free(ones); free(rhs); free(x);
  return fail ? 1 : 0;
}
