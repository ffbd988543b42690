"""Seeded fuzz of the HIP factor + solve path over matrix shapes and tile / panel /
amalgamation options that divide nothing evenly (nb 5..200, panel width 4..64,
nemin 1..64, engine variants), each against the CPU oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import bwd_err, lower_mask, make_case, oracle_factor, rel_err
from spllt_amd import matgen

pytestmark = pytest.mark.gpu


def _random_spd(rng, n, density):
    M = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)),
                  format="csr")
    S = (M + M.T).tocsr()
    S.data[:] = -np.abs(S.data)
    S.setdiag(0)
    S.eliminate_zeros()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0
    return (S + sp.diags(d)).tocsc()


@pytest.mark.parametrize("seed", range(64))
def test_fuzz_factor_and_solve(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = seed % 4
    if kind == 0:
        A = matgen.nd_like(tuple(int(x) for x in rng.integers(4, 11, size=3)), int(rng.integers(1, 3)))
    elif kind == 1:
        A = matgen.poisson2d(int(rng.integers(5, 40)))
    elif kind == 2:
        A = _random_spd(rng, int(rng.integers(20, 400)), float(rng.uniform(0.01, 0.2)))
    else:
        A = matgen.fe27(tuple(int(x) for x in rng.integers(3, 7, size=3)), int(rng.integers(1, 4)))
    nb = int(rng.choice([5, 7, 16, 24, 33, 48, 64, 100, 130, 200]))
    pw = int(rng.choice([4, 5, 8, 10, 12, 16, 24, 32, 40, 48, 64]))
    nemin = int(rng.choice([1, 4, 16, 32, 64]))
    flags = int(rng.choice([0, 0, 0, 2, 4, 12, 16, 32, 64]))
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw, engine_flags=flags)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) <= 1e-12, (nb, pw, nemin, flags)
    assert np.all(got[~mask] == 0.0)
    nrhs = int(rng.integers(1, 6))
    X = rng.standard_normal((f.n, nrhs))
    B = A @ X
    Y = f.solve(B)
    for q in range(nrhs):
        assert bwd_err(A, Y[:, q], B[:, q]) <= 1e-14, (nb, pw, nemin, flags)


@pytest.mark.parametrize("flags", [0, 4, 12, 16, 32, 64])
@pytest.mark.parametrize("nb,pw", [(48, 5), (48, 24), (100, 10), (100, 40), (130, 48), (33, 12)])
def test_ragged_panels_in_every_engine_variant(flags, nb, pw):
    """panel widths that are no multiple of 16 (or 4) and do not divide nb, in
    every engine variant (the fused strip kernel once zeroed the columns of the
    next panel when a panel ended inside a 16-column MFMA tile)"""
    A = matgen.fe27((5, 4, 4), 3)
    f, val = make_case(A, nb=nb, nemin=4, panel_width=pw, engine_flags=flags)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(f)) <= 1e-12
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14
