"""Shared helpers of the test-suite (CPU side)."""
import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp

from oracle import pyoracle
from spllt_amd import api, matgen  # noqa: F401


def sym_tables(f):
    return {k: f.sym(k) for k in ("order", "sptr", "sparent", "rptr", "rlist", "bcol_off",
                                  "bcol_width", "bcol_r0", "bcol_nrow", "bcol_node", "node_bcol0",
                                  "level", "small")}


def lower_mask(f):
    """Boolean mask over the arena: True on entries that hold L (row >= col);
    False on the never-read strict upper triangle of diagonal tiles
    (parity checks must mask it)."""
    t = sym_tables(f)
    mask = np.zeros(f.sym_info()["arena"], dtype=bool)
    for b in range(len(t["bcol_off"])):
        w, nr, off = int(t["bcol_width"][b]), int(t["bcol_nrow"][b]), int(t["bcol_off"][b])
        m = np.ones((nr, w), dtype=bool)
        m[:w, :w] = np.tril(np.ones((w, w), dtype=bool))
        mask[off:off + nr * w] = m.ravel()
    return mask


def dense_arena(f, A):
    """Expected arena from an independent dense LAPACK Cholesky of P A P^T."""
    t = sym_tables(f)
    n = f.n
    P = np.empty(n, dtype=np.int64)
    P[t["order"]] = np.arange(n)
    Ld = sl.cholesky(sp.csc_matrix(A).toarray()[np.ix_(P, P)], lower=True)
    arena = np.zeros(f.sym_info()["arena"])
    for b in range(len(t["bcol_off"])):
        s = int(t["bcol_node"][b])
        rows = t["rlist"][t["rptr"][s]:t["rptr"][s + 1]]
        w, nr, off, r0 = (int(t["bcol_width"][b]), int(t["bcol_nrow"][b]), int(t["bcol_off"][b]),
                          int(t["bcol_r0"][b]))
        c0 = int(t["sptr"][s]) + r0
        blk = Ld[np.ix_(rows[r0:r0 + nr], np.arange(c0, c0 + w))]
        arena[off:off + nr * w] = blk.ravel()
    return arena


def make_case(A, nb, nemin=32, prune=False, ncpu=1, order=None, **kw):
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=nb, nemin=nemin, prune_tree=prune, ncpu=ncpu,
                          order=order, **kw)
    return f, val


def oracle_factor(f, val, variant="plain", nthreads=1, use_small=False, min_width_blas=8):
    small = f.sym("small") if use_small else None
    o = pyoracle.OracleFactor.from_factorization(f, small=small, variant=variant,
                                                 min_width_blas=min_width_blas)
    rc = o.factorize(val, nthreads)
    return o, rc


def rel_err(a, b, mask=None):
    if mask is not None:
        a, b = a[mask], b[mask]
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def bwd_err(A, x, b):
    """Scaled backward error of the reference's checker
    (check_backward_error_multi, src/utils_mod.F90:432-478):
    ||b - A x||_2 / (||b||_2 + max|a_ij| ||x||_2); pass iff <= 1e-14."""
    r = b - A @ x
    return float(np.linalg.norm(r) / (np.linalg.norm(b) + abs(A).max() * np.linalg.norm(x)))
