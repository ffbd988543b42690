"""The substitution program (level-batched diag / strip launches, partition-aware)
interpreted in numpy: equals a dense triangular solve, and its three-phase
partitioned form with two sums of the right-hand-side vector (what the
all-reduces do) reproduces the same solution for 2..4 ranks."""
import numpy as np
import pytest
import scipy.sparse as sp

from spllt_amd import matgen
from helpers import dense_arena, make_case
from emulate import emulate_solve

GENS = [lambda: matgen.nd_like((8, 7, 7), 2), lambda: matgen.poisson2d(24),
        lambda: sp.block_diag([matgen.poisson2d(6), matgen.poisson2d(5)]).tocsc()]


@pytest.mark.parametrize("gen", GENS)
@pytest.mark.parametrize("nb,pw", [(16, 8), (48, 16), (200, 64)])
def test_solve_program_equals_dense_solve(gen, nb, pw):
    A = gen()
    f, val = make_case(A, nb=nb, nemin=8, panel_width=pw)
    L = dense_arena(f, A)
    n = f.n
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, 3))
    B = A @ X
    pos = f.sym("order")
    Y = np.zeros((3, n))
    Y[:, pos] = B.T
    emulate_solve(f, L, Y)
    np.testing.assert_allclose(Y[:, pos].T, X, rtol=0, atol=1e-9)
    # forward then backward separately = both at once
    Y2 = np.zeros((3, n))
    Y2[:, pos] = B.T
    emulate_solve(f, L, Y2, job=1)
    emulate_solve(f, L, Y2, job=2)
    np.testing.assert_allclose(Y2, Y, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_partitioned_solve_program(world):
    A = matgen.nd_like((9, 8, 8), 2)
    fs = []
    for r in range(world):
        f, val = make_case(A, nb=32, nemin=8, prune=True, ncpu=world, panel_width=16)
        f.set_partition(r, world)
        fs.append(f)
    L = dense_arena(fs[0], A)          # every rank may read the whole factor here; it only
    n = fs[0].n                        # touches its own branches and the top tree
    owner, sptr, pos = fs[0].partition("owner"), fs[0].sym("sptr"), fs[0].sym("order")
    own = owner[np.repeat(np.arange(len(sptr) - 1), np.diff(sptr))]
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, 2))
    B = A @ X
    ys, masks = [], []
    for r in range(world):
        m = (own == r) | ((own < 0) & (r == 0))
        Y = np.zeros((2, n))
        Y[:, pos] = B.T
        ys.append(Y * m)
        masks.append(m)
    assert sum(m.sum() for m in masks) == n
    for f, y in zip(fs, ys):
        emulate_solve(f, L, y, phase=0)
    total = np.sum(ys, axis=0)
    ys = [total.copy() for _ in ys]
    for f, y, m in zip(fs, ys, masks):
        emulate_solve(f, L, y, phase=1)
        emulate_solve(f, L, y, phase=2)
        y *= m
    got = np.sum(ys, axis=0)[:, pos].T
    np.testing.assert_allclose(got, X, rtol=0, atol=1e-9)
    # the phases of one rank only touch its own branches and the top tree
    nsub, ntop = fs[1].program("solve_split")
    assert 0 < nsub < len(fs[1].program("solve_fwd")) and 0 < ntop < len(fs[1].program("solve_bwd"))
