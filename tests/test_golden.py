"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
dense LAPACK Cholesky factors of P A P^T for a FIXED user order.  The oracle
(CPU) and the numpy-interpreted program must reproduce them; the GPU path is
checked against the same files in test_gpu_parity.py::test_golden_vectors_gpu."""
import glob
import os

import numpy as np
import pytest

from emulate import emulate_program
from helpers import lower_mask, oracle_factor, sym_tables
from spllt_amd import api

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dense_chol_*.npz")))


def expected_arena(f, g):
    t = sym_tables(f)
    n = f.n
    pos_fix = g["order_in"].astype(np.int64) - 1       # var -> fixture position
    var_of_pos = np.empty(n, dtype=np.int64)
    var_of_pos[t["order"]] = np.arange(n)               # product position -> var
    fixpos = pos_fix[var_of_pos]                        # product position -> fixture position
    # L (product order) = Q L_fix-structure only if both orders give the same factor up to a
    # symmetric permutation that keeps it triangular; the product only applies an etree
    # postorder + supernode amalgamation, which does: L_prod = Pi L_fix Pi^T.
    Ld = g["L_dense"]
    Lp = Ld[np.ix_(fixpos, fixpos)]
    arena = np.zeros(f.sym_info()["arena"])
    for b in range(len(t["bcol_off"])):
        s = int(t["bcol_node"][b])
        rows = t["rlist"][t["rptr"][s]:t["rptr"][s + 1]]
        w, nr, off, r0 = (int(t["bcol_width"][b]), int(t["bcol_nrow"][b]), int(t["bcol_off"][b]),
                          int(t["bcol_r0"][b]))
        c0 = int(t["sptr"][s]) + r0
        arena[off:off + nr * w] = Lp[np.ix_(rows[r0:r0 + nr], np.arange(c0, c0 + w))].ravel()
    return arena, Lp


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
@pytest.mark.parametrize("nb", [4, 16])
def test_oracle_and_program_reproduce_golden(path, nb):
    g = np.load(path)
    f = api.Factorization(int(g["n"]), g["ptr"], g["row"], nb=nb, nemin=4, prune_tree=False,
                          order=g["order_in"], panel_width=16)
    exp, Lp = expected_arena(f, g)
    assert np.allclose(np.triu(Lp, 1), 0, atol=1e-13), "postorder kept L triangular"
    mask = lower_mask(f)
    o, rc = oracle_factor(f, g["val"])
    assert rc == 0
    assert np.abs(o.arena() - exp)[mask].max() <= 2e-14 * np.abs(exp).max()
    np.testing.assert_allclose(o.solve(g["b"]), g["x"], rtol=0, atol=1e-12)
    got = emulate_program(f, g["val"])
    assert np.abs(got - exp)[mask].max() <= 1e-13 * np.abs(exp).max()
