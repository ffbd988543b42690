"""Seeded fuzz of the scheduler on the CPU: random shapes, tile / panel /
amalgamation options and engine variants; the exported program, interpreted in
numpy (tests/emulate.py), must reproduce the dense Cholesky factor, and the
stream DAG must order every conflict.  Partitioned programs (2..4 ranks) go
through the same check with a numpy sum as the exchange."""
import numpy as np
import pytest
import scipy.sparse as sp

from spllt_amd import matgen
from helpers import dense_arena, lower_mask, make_case, rel_err
from emulate import emulate_program, emulate_ranks
import test_schedule as ts


def _random_spd(rng, n, density):
    M = sp.random(n, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                  format="csr")
    S = (M + M.T).tocsr()
    S.data[:] = -np.abs(S.data)
    S.setdiag(0)
    S.eliminate_zeros()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0
    return (S + sp.diags(d)).tocsc()


def _matrix(rng, seed):
    kind = seed % 3
    if kind == 0:
        return matgen.nd_like(tuple(int(x) for x in rng.integers(4, 9, size=3)), int(rng.integers(1, 3)))
    if kind == 1:
        return matgen.poisson2d(int(rng.integers(5, 28)))
    return _random_spd(rng, int(rng.integers(20, 220)), float(rng.uniform(0.01, 0.15)))


def _dag_is_ordered(f):
    return not ts.dag_violations(f)[0]


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_program_single_gpu(seed, monkeypatch):
    rng = np.random.default_rng(3000 + seed)
    A = _matrix(rng, seed)
    nb = int(rng.choice([5, 8, 16, 24, 33, 48, 100]))
    pw = int(rng.choice([4, 5, 8, 12, 16, 24, 64]))
    nemin = int(rng.choice([1, 4, 16, 32]))
    flags = int(rng.choice([0, 0, 2, 64, 66, 1024, 2048, 4096, 4098]))
    cb = int(rng.choice([0, 0, 16, 24, 40, 64]))
    if cb:
        monkeypatch.setenv("SPLLT_CHAIN_BLOCK", str(cb))
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw, engine_flags=flags)
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-12, (nb, pw, nemin, flags, cb)
    assert _dag_is_ordered(f), (nb, pw, nemin, flags, cb)


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_program_partitioned(seed):
    rng = np.random.default_rng(4000 + seed)
    A = _matrix(rng, seed)
    world = int(rng.integers(2, 5))
    nb = int(rng.choice([8, 16, 32, 48]))
    pw = int(rng.choice([8, 16, 24]))
    # top tree replicated on every rank / distributed over the ranks (owner computes)
    flags = int(rng.choice([8192, 16384])) | int(rng.choice([0, 0, 2, 64, 512, 1024, 2048, 4096]))   # ... in any engine variant
    fs, vals = [], None
    for r in range(world):
        f, vals = make_case(A, nb=nb, nemin=8, prune=True, ncpu=world, panel_width=pw, engine_flags=flags)
        f.set_partition(r, world)
        fs.append(f)
    ref, mask = dense_arena(fs[0], A), lower_mask(fs[0])
    owner, bc_node = fs[0].partition("owner"), fs[0].sym("bcol_node")
    off, w, nr = fs[0].sym("bcol_off"), fs[0].sym("bcol_width"), fs[0].sym("bcol_nrow")
    arenas = emulate_ranks(fs, vals)     # all ranks in lockstep, numpy collectives
    for r, f in enumerate(fs):
        got = arenas[r]
        mine = np.zeros_like(mask)
        for b in range(len(off)):
            if owner[bc_node[b]] in (r, -1):
                mine[off[b]:off[b] + nr[b] * w[b]] = True
        assert rel_err(got, ref, mask & mine) < 1e-12, (world, nb, pw, r, flags)
        assert _dag_is_ordered(f), (world, nb, pw, r, flags)
