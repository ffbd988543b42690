"""Degenerate and ragged inputs of the factor path: order-1 and order-2 matrices,
a diagonal matrix (every supernode a leaf and a root), a disconnected graph (an
elimination forest), arrow matrices (one dense row first / last), a dense
matrix with tile sizes that divide nothing, nb larger than n, n = 0.  The CPU
half interprets the exported program in numpy (tests/emulate.py) against a dense
Cholesky; the GPU half runs the HIP engine through the C-ABI against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from spllt_amd import api, matgen
from helpers import make_case, dense_arena, lower_mask, rel_err, oracle_factor, bwd_err
from emulate import emulate_program


def _arrow(n, first):
    A = sp.lil_matrix((n, n))
    A.setdiag(n + 1.0)
    k = 0 if first else n - 1
    A[k, :] = 1.0
    A[:, k] = 1.0
    A[k, k] = 2.0 * n
    return A.tocsc()


EDGE = [
    ("n1", lambda: sp.csc_matrix(np.array([[4.0]])), dict(nb=8, nemin=4)),
    ("n2-dense", lambda: sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 3.0]])), dict(nb=8, nemin=4)),
    ("diagonal", lambda: sp.diags([np.arange(1, 41.0)], [0]).tocsc(), dict(nb=8, nemin=4)),
    ("forest", lambda: sp.block_diag([matgen.poisson2d(6), matgen.poisson2d(5),
                                      sp.csc_matrix(np.array([[2.0]]))]).tocsc(), dict(nb=8, nemin=4)),
    ("arrow-last", lambda: _arrow(60, False), dict(nb=16, nemin=4)),
    ("arrow-first", lambda: _arrow(60, True), dict(nb=16, nemin=4)),
    ("dense40-nb16", lambda: sp.csc_matrix(np.ones((40, 40)) + 40 * np.eye(40)), dict(nb=16, nemin=4)),
    ("dense40-nb7-pw5", lambda: sp.csc_matrix(np.ones((40, 40)) + 40 * np.eye(40)),
     dict(nb=7, nemin=4, panel_width=5)),
    ("dense130-nb64-pw48", lambda: sp.csc_matrix(np.ones((130, 130)) + 130 * np.eye(130)),
     dict(nb=64, nemin=4, panel_width=48)),
    ("nb-larger-than-n", lambda: matgen.poisson2d(7), dict(nb=1000, nemin=64)),
]
# the widest supported block column (nb = 1024) with a ragged second one; GPU only (the numpy
# interpreter would take a while on 1100^3)
EDGE_GPU = EDGE + [("dense1100-nb1024", lambda: sp.csc_matrix(np.ones((1100, 1100)) + 1100 * np.eye(1100)),
                    dict(nb=1024, nemin=4))]


@pytest.mark.parametrize("name,gen,kw", EDGE, ids=[e[0] for e in EDGE])
@pytest.mark.parametrize("flags", [0, 2, 64])
def test_edge_case_program_reproduces_dense_cholesky(name, gen, kw, flags):
    A = gen()
    f, val = make_case(A, engine_flags=flags, **kw)
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


def test_empty_matrix_is_accepted():
    f = api.Factorization(0, np.array([1], dtype=np.int32), np.array([], dtype=np.int32), nb=8)
    si = f.sym_info()
    assert si["n"] == 0 and si["nnodes"] == 0 and si["arena"] == 0
    assert len(f.program("launches")) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,gen,kw", EDGE_GPU, ids=[e[0] for e in EDGE_GPU])
def test_edge_case_factor_and_solve_on_gpu(name, gen, kw):
    A = gen()
    f, val = make_case(A, **kw)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) <= 1e-12
    assert np.all(got[~mask] == 0.0)
    b = A @ np.ones(f.n)
    x = f.solve(b)
    assert bwd_err(A, x, b) <= 1e-14


@pytest.mark.gpu
def test_empty_matrix_on_gpu():
    f = api.Factorization(0, np.array([1], dtype=np.int32), np.array([], dtype=np.int32), nb=8)
    f.factor(np.zeros(0)).wait()
    assert f.get_factor().size == 0
