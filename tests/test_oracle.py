"""CPU tests that pin the oracle (oracle/) before it is trusted as the checker.

Pins: (1) the reference's only known-answer case for this path,
example/C/simple.c:25-75 (3x3 tridiag(-1,2,-1), nb=4): L and x=(1.5,2,1.5);
(2) an independent dense LAPACK Cholesky of P A P^T on small Poisson / box
stencil matrices (the Cholesky factor is unique, so any correct restatement
must agree to rounding); (3) the reference's residual bar
||Ax-b||/||b|| <= 1e-14 (drivers/spllt_omp_bench.F90:389).
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import bwd_err, dense_arena, lower_mask, make_case, oracle_factor, rel_err
from spllt_amd import matgen

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_kat_simple_c():
    """example/C/simple.c through analyse + oracle factor + oracle solve."""
    gold = json.load(open(os.path.join(GOLD, "kat_simple_c.json")))
    A = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(3, 3)).tocsc()
    f, val = make_case(A, nb=4)
    assert list(f.sym("order")) == [0, 1, 2]
    o, rc = oracle_factor(f, val)
    assert rc == 0
    np.testing.assert_allclose(o.arena(), gold["L_rowmajor"], rtol=0, atol=5e-9)
    np.testing.assert_allclose(o.solve(np.ones(3)), gold["x"], rtol=1e-15, atol=1e-15)


CASES = [
    ("p2d8", lambda: matgen.poisson2d(8), 4, 4),
    ("p2d16", lambda: matgen.poisson2d(16), 8, 4),
    ("p2d16b", lambda: matgen.poisson2d(16), 16, 8),
    ("p2d32", lambda: matgen.poisson2d(32), 16, 32),
    ("p3d6", lambda: matgen.poisson3d(6), 8, 4),
    ("box6", lambda: matgen.nd_like((6, 6, 6), 2), 16, 8),
]


@pytest.mark.parametrize("name,gen,nb,nemin", CASES)
@pytest.mark.parametrize("prune,ncpu", [(False, 1), (True, 1), (True, 2), (True, 4)])
@pytest.mark.parametrize("variant", ["plain", "mkl"])
def test_oracle_vs_dense_cholesky(name, gen, nb, nemin, prune, ncpu, variant):
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin, prune=prune, ncpu=ncpu)
    try:
        o, rc = oracle_factor(f, val, variant=variant, nthreads=1 if variant == "plain" else 3,
                              use_small=prune)
    except RuntimeError:
        pytest.skip("MKL build of the oracle not available")
    assert rc == 0
    exp = dense_arena(f, A)
    assert rel_err(o.arena(), exp, lower_mask(f)) < 2e-14
    b = A @ np.ones(f.n)
    x = o.solve(b)
    assert bwd_err(A, x, b) <= 1e-14


def test_oracle_update_direct_path():
    """n1 < min_width_blas routes through spllt_update_direct (kernels_mod:14-93)."""
    A = matgen.poisson2d(16)
    f, val = make_case(A, nb=8, nemin=4)
    o, rc = oracle_factor(f, val, min_width_blas=1000)
    assert rc == 0
    assert rel_err(o.arena(), dense_arena(f, A), lower_mask(f)) < 2e-14


def test_oracle_not_posdef_reported():
    A = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(6, 6)).tolil()
    A[3, 3] = -5.0
    f, val = make_case(A.tocsc(), nb=4, nemin=1)
    o, rc = oracle_factor(f, val)
    assert rc > 0


def test_oracle_tiling_matches_product_symbolic():
    """spo_create (restating analyse_mod:305-469,1033-1171) and the product's
    analyse must agree on tiles and on the val->lcol map."""
    A = matgen.poisson2d(24)
    f, val = make_case(A, nb=8, nemin=8)
    o, _ = oracle_factor(f, val)
    info = f.sym_info()
    assert o.lib.spo_nbcol(o.h) == info["nbcol"]
    assert o.lib.spo_nblk(o.h) == info["nblk"]
    assert o.lib.spo_arena(o.h) == info["arena"]
    assert o.lib.spo_maxmn(o.h) == info["maxmn"]
    off = f.sym("bcol_off")
    lp = f.sym("lmap_ptr")
    md, ms = f.sym("map_dst"), f.sym("map_src")
    for b in range(info["nbcol"]):
        d, s = o.lmap(b)
        mine = sorted(zip((md[lp[b]:lp[b + 1]] - off[b]).tolist(), ms[lp[b]:lp[b + 1]].tolist()))
        assert mine == sorted(zip(d.tolist(), s.tolist()))
    # tile descriptors: contiguous per block column, diagonal first
    w, nr = f.sym("bcol_width"), f.sym("bcol_nrow")
    blocks = o.blocks()
    k = 0
    for b in range(info["nbcol"]):
        ntile = (nr[b] - 1) // info["nb"] + 1
        for t in range(ntile):
            bid, dblk, last, sa, bc, blkm, blkn, _ = blocks[k]
            assert (bc, blkn, dblk, last) == (b, w[b], k - t, k - t + ntile - 1)
            assert blkm == min(info["nb"], nr[b] - t * info["nb"])
            assert sa == t * info["nb"] * w[b]
            k += 1
