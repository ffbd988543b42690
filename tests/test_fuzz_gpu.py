"""Seeded fuzz of the HIP factor + solve path over matrix shapes and tile / panel /
amalgamation options that divide nothing evenly (nb 5..200, panel width 4..64,
nemin 1..64, engine variants), each against the CPU oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import bwd_err, drive_exchanges, lower_mask, make_case, oracle_factor, rel_err
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


@pytest.mark.parametrize("seed", range(int(os.environ.get("SPLLT_FUZZ_SEEDS", "64"))))    # (a longer campaign: SPLLT_FUZZ_SEEDS=1024)
def test_fuzz_factor_and_solve(seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    # every third case with chain blocks of four panels (k_chain_block + k_trsm_rows: ragged panel
    # widths that are no multiple of 4, block columns of 1-40 panels), the graph replays and eager launches in turn
    monkeypatch.setenv("SPLLT_CHAIN4", "1" if seed % 3 == 2 else "0")
    monkeypatch.setenv("SPLLT_HIP_GRAPH", str(seed % 4 - 1) if seed % 4 else "-1")
    # every other case with the small subtrees as single device tasks (k_subtree), at budgets that make
    # tasks of one node, of a few, and of whole trees
    monkeypatch.setenv("SPLLT_SUBTREES", str(seed // 2 % 2))
    monkeypatch.setenv("SPLLT_SUBTREE_US", str([40, 300, 5000][seed % 3]))
    # (the chain replay of a small factorization: the single-stream program, the library's default, or the
    # multi-stream program in program order)
    monkeypatch.setenv("SPLLT_CHAIN_GRAPH_SERIAL", str(seed // 4 % 2))
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
    flags = int(rng.choice([0, 0, 0, 2, 64, 66, 256, 320, 1024, 1088, 2048, 4096]))
    cb = int(rng.choice([0, 0, 32, 48, 96, 128, 256]))
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw, engine_flags=flags, chain_block=cb or None)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) <= 1e-12, (nb, pw, nemin, flags, cb)
    assert np.all(got[~mask] == 0.0)
    nrhs = int(rng.integers(1, 6))
    X = rng.standard_normal((f.n, nrhs))
    B = A @ X
    Y = f.solve(B)
    for q in range(nrhs):
        assert bwd_err(A, Y[:, q], B[:, q]) <= 1e-14, (nb, pw, nemin, flags)


@pytest.mark.parametrize("flags", [0, 2, 64, 2048])
@pytest.mark.parametrize("cb", [None, 40, 256])
@pytest.mark.parametrize("nb,pw", [(48, 5), (48, 24), (100, 10), (100, 40), (130, 48), (33, 12)])
@pytest.mark.parametrize("chain4", [0, 1])
def test_ragged_panels_in_every_engine_variant(flags, cb, nb, pw, chain4, monkeypatch):
    """panel widths that are no multiple of 16 (or 4) and do not divide nb, in
    every engine variant and with sub-tiles that end inside a block column; with one panel per
    chain step and with chain blocks of four panels (k_chain_block / k_trsm_rows on 5-, 10-, 12-,
    24-, 40- and 48-wide panels)"""
    monkeypatch.setenv("SPLLT_CHAIN4", str(chain4))
    A = matgen.fe27((5, 4, 4), 3)
    f, val = make_case(A, nb=nb, nemin=4, panel_width=pw, engine_flags=flags, chain_block=cb)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(f)) <= 1e-12
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14


def _partitioned_factor_and_solve(A, world, nb, nemin, pw, flags=0):
    """`world` rank-engines on this device; torch sums stand for the all-reduces."""
    import torch
    fs, bufs = [], []
    for r in range(world):
        f, val = make_case(A, nb=nb, nemin=nemin, prune=True, ncpu=world, panel_width=pw, engine_flags=flags)
        xel = f.set_partition(r, world)
        xb = torch.zeros(max(xel, 1), dtype=torch.float64, device="cuda")
        f.set_exchange_buffer(xb.data_ptr())
        fs.append(f)
        bufs.append(xb)
    dval = torch.tensor(val, device="cuda")
    for f in fs:
        f.factor_dev(dval.data_ptr())
    drive_exchanges(fs, bufs)
    for f in fs:
        f.wait()
    n = fs[0].n
    owner, sptr, pos = fs[0].partition("owner"), fs[0].sym("sptr"), fs[0].sym("order")
    own = owner[np.repeat(np.arange(len(sptr) - 1), np.diff(sptr))]
    rng = np.random.default_rng(7)
    X = rng.standard_normal((n, 2))
    B = A @ X
    ys, masks = [], []
    for r in range(world):
        m = torch.tensor((own == r) | ((own < 0) & (r == 0)), device="cuda")
        Y = np.zeros((2, n))
        Y[:, pos] = B.T
        ys.append(torch.tensor(Y, device="cuda") * m)
        masks.append(m)
    torch.cuda.synchronize()
    for f, y in zip(fs, ys):
        f.solve_dev(y.data_ptr(), 2, 0, 0)
    total = torch.stack(ys).sum(dim=0)
    for y in ys:
        y.copy_(total)
    torch.cuda.synchronize()   # torch's stream is not the engines' stream
    for f, y, m in zip(fs, ys, masks):
        f.solve_dev(y.data_ptr(), 2, 0, 1)
        f.solve_dev(y.data_ptr(), 2, 0, 2)
        y *= m
    got = torch.stack(ys).sum(dim=0).cpu().numpy()[:, pos].T
    return fs, val, got, B


@pytest.mark.parametrize("seed", range(int(os.environ.get("SPLLT_FUZZ_PART_SEEDS", "16"))))
def test_fuzz_partitioned_factor_and_solve(seed, monkeypatch):
    """subtree partition over 2..5 ranks (some may own nothing, the top tree may be
    one node) on random shapes: L of every rank against the oracle on the block
    columns it holds, the partitioned solve against the residual bar; every other case with chain
    blocks of four panels (the distributed top tree's owners run k_chain_block / k_trsm_rows)"""
    monkeypatch.setenv("SPLLT_CHAIN4", str(seed % 2))
    rng = np.random.default_rng(2000 + seed)
    if seed % 3 == 0:
        A = matgen.nd_like(tuple(int(x) for x in rng.integers(5, 11, size=3)), int(rng.integers(1, 3)))
    elif seed % 3 == 1:
        A = matgen.poisson2d(int(rng.integers(8, 40)))
    else:
        A = _random_spd(rng, int(rng.integers(40, 300)), float(rng.uniform(0.01, 0.1)))
    world = int(rng.integers(2, 6))
    nb = int(rng.choice([8, 16, 32, 48, 100]))
    pw = int(rng.choice([8, 16, 24, 64]))
    nemin = int(rng.choice([4, 16, 32]))
    # top tree distributed over the ranks / replicated, in any engine variant
    flags = int(rng.choice([8192, 16384])) | int(rng.choice([0, 0, 2, 64, 512, 1024, 2048, 4096]))
    fs, val, got, B = _partitioned_factor_and_solve(A, world, nb, nemin, pw, flags)
    o, rc = oracle_factor(fs[0], val)
    assert rc == 0
    ref, mask = o.arena(), lower_mask(fs[0])
    owner, bc_node = fs[0].partition("owner"), fs[0].sym("bcol_node")
    off, w, nr = fs[0].sym("bcol_off"), fs[0].sym("bcol_width"), fs[0].sym("bcol_nrow")
    for r, f in enumerate(fs):
        L = f.get_factor()
        mine = np.zeros_like(mask)
        for b in range(len(off)):
            if owner[bc_node[b]] in (r, -1):
                mine[off[b]:off[b] + nr[b] * w[b]] = True
        assert rel_err(L, ref, mask & mine) <= 1e-12, (world, nb, pw, nemin, r)
    for q in range(2):
        assert bwd_err(A, got[:, q], B[:, q]) <= 1e-14, (world, nb, pw, nemin)
