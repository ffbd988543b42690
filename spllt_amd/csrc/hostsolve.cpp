// Host-side triangular solves with the tiled factor (CPU, out of the hot
// path; SURVEY.md section 8(f) row f2 is the GPU version).  Restates the math
// of solve_fwd / solve_bwd (reference src/spllt_solve_mod.F90:244-411):
//   job 1:  y = L^-1 P b      job 2:  x = P^T L^-T y      job 0: both.
#include "hostsolve.hpp"

#include <vector>

namespace spx {

void host_solve(const Symbolic& S, const double* L, int nrhs, double* x, int job) {
  const int n = S.n;
  std::vector<double> y(n);
  for (int r = 0; r < nrhs; ++r) {
    double* xr = x + (int64_t)r * n;
    for (int i = 0; i < n; ++i) y[S.order[i]] = xr[i];
    if (job == 0 || job == 1) {
      for (int s = 0; s < S.nnodes; ++s) {
        const int* idx = S.rows(s);
        for (int b = S.node_bcol0[s]; b < S.node_bcol0[s + 1]; ++b) {
          const BlockCol& B = S.bcols[b];
          const double* lc = L + B.off;
          const int w = B.width;
          for (int j = 0; j < w; ++j) {
            double v = y[idx[B.r0 + j]] / lc[(int64_t)j * w + j];
            y[idx[B.r0 + j]] = v;
            for (int i = j + 1; i < B.nrow; ++i) y[idx[B.r0 + i]] -= lc[(int64_t)i * w + j] * v;
          }
        }
      }
    }
    if (job == 0 || job == 2) {
      for (int s = S.nnodes - 1; s >= 0; --s) {
        const int* idx = S.rows(s);
        for (int b = S.node_bcol0[s + 1] - 1; b >= S.node_bcol0[s]; --b) {
          const BlockCol& B = S.bcols[b];
          const double* lc = L + B.off;
          const int w = B.width;
          for (int j = w - 1; j >= 0; --j) {
            double v = y[idx[B.r0 + j]];
            for (int i = j + 1; i < B.nrow; ++i) v -= lc[(int64_t)i * w + j] * y[idx[B.r0 + i]];
            y[idx[B.r0 + j]] = v / lc[(int64_t)j * w + j];
          }
        }
      }
    }
    for (int i = 0; i < n; ++i) xr[i] = y[S.order[i]];
  }
}

}  // namespace spx
