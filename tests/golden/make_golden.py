#!/usr/bin/env python3
"""Generates the committed golden vectors of tests/golden/ (run from the repo root:
`python tests/golden/make_golden.py`).

* kat_simple_c.json is written by hand from reference example/C/simple.c:25-75
  (see its "source" field) and is not regenerated here.
* dense_chol_*.npz: small SPD matrices in the C-ABI's 1-based CSC-lower form, a
  fixed pivot order (geometric nested dissection, so that the fixture does not
  depend on the built-in ordering), and the exact answer the factorize path must
  reproduce: the dense LAPACK Cholesky factor of P A P^T (scipy.linalg.cholesky)
  and x = A^-1 b for b = A 1.  These are independent known answers (the
  Cholesky factor is unique); neither the oracle nor the HIP path is involved
  in producing them.
"""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spllt_amd import api, matgen  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "dense_chol_p2d8": (lambda: matgen.poisson2d(8), (8, 8), 1),
    "dense_chol_p3d5": (lambda: matgen.poisson3d(5), (5, 5, 5), 1),
    "dense_chol_box5": (lambda: matgen.nd_like((5, 5, 4), 2), (5, 5, 4), 2),
}
for name, (gen, shape, radius) in CASES.items():
    A = gen()
    n, ptr, row, val = api.csc_lower_1based(A)
    order = matgen.geometric_nd_order(shape, radius, leaf=4)  # 1-based positions
    P = np.empty(n, dtype=np.int64)
    P[order - 1] = np.arange(n)
    Ld = sl.cholesky(A.toarray()[np.ix_(P, P)], lower=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), n=n, ptr=ptr, row=row, val=val,
                        order_in=order, L_dense=Ld, b=A @ np.ones(n), x=np.ones(n))
    print(name, n, "nnz", len(val))
