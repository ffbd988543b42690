#pragma once
#include "symbolic.hpp"

namespace spx {
// x (n x nrhs, column-major, original variable order) is overwritten.
void host_solve(const Symbolic& S, const double* L, int nrhs, double* x, int job);
}  // namespace spx
