"""CPU tests of the host side of the product: analyse stand-in (orderings,
supernodes, tiling) and the stream-DAG program (work tables), the latter by
interpreting the exported tables in numpy (tests/emulate.py) and comparing
with the oracle / a dense Cholesky."""
import numpy as np
import pytest

from emulate import emulate_program
from helpers import dense_arena, lower_mask, make_case, oracle_factor, rel_err
from spllt_amd import matgen

CASES = [
    ("p2d12-nb4", lambda: matgen.poisson2d(12), 4, 4, 64),
    ("p2d16-nb8", lambda: matgen.poisson2d(16), 8, 4, 64),
    ("p2d24-nb200", lambda: matgen.poisson2d(24), 200, 32, 64),
    ("p3d6-nb8", lambda: matgen.poisson3d(6), 8, 4, 64),
    ("box6-nb96-pw32", lambda: matgen.nd_like((6, 6, 6), 2), 96, 8, 32),
    ("box7-nb256", lambda: matgen.nd_like((7, 7, 6), 2), 256, 16, 64),
]


@pytest.mark.parametrize("name,gen,nb,nemin,pw", CASES)
def test_program_tables_reproduce_oracle(name, gen, nb, nemin, pw):
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw)
    got = emulate_program(f, val)
    o, rc = oracle_factor(f, val)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) < 1e-13
    assert rel_err(got, dense_arena(f, A), mask) < 1e-13
    # strictly-upper triangle of diagonal tiles is never written by the program
    assert np.all(got[~mask] == 0.0)


@pytest.mark.parametrize("gen", [lambda: matgen.poisson2d(20), lambda: matgen.poisson3d(7),
                                 lambda: matgen.nd_like((8, 7, 6), 2),
                                 lambda: matgen.fe27((4, 4, 3), 3)])
@pytest.mark.parametrize("use_geo", [False, True])
def test_symbolic_structure_is_valid(gen, use_geo):
    """order is a permutation, nodes are postordered, row lists are sorted,
    own columns first, and contain the pattern of A and of the children."""
    A = gen()
    order = None
    f, val = make_case(A, nb=16, nemin=8, order=order)
    n = f.n
    o = f.sym("order")
    assert sorted(o.tolist()) == list(range(n))
    sptr, sparent, rptr, rlist = f.sym("sptr"), f.sym("sparent"), f.sym("rptr"), f.sym("rlist")
    nn = len(sparent)
    assert sptr[0] == 0 and sptr[-1] == n
    Ap = A.tocsc()
    for s in range(nn):
        assert sparent[s] > s
        rows = rlist[rptr[s]:rptr[s + 1]]
        nc = sptr[s + 1] - sptr[s]
        assert (rows[:nc] == np.arange(sptr[s], sptr[s + 1])).all()
        assert (np.diff(rows) > 0).all()
        p = sparent[s]
        if p < nn:
            prow = set(rlist[rptr[p]:rptr[p + 1]].tolist())
            assert set(rows[nc:].tolist()) <= prow
    inv = np.empty(n, dtype=np.int64)
    inv[o] = np.arange(n)
    snode = np.repeat(np.arange(nn), np.diff(sptr))
    for j in range(n):
        for i in Ap.indices[Ap.indptr[j]:Ap.indptr[j + 1]]:
            a, b = sorted((o[i], o[j]))
            s = snode[a]
            assert b in set(rlist[rptr[s]:rptr[s + 1]].tolist())
    info = f.sym_info()
    m = np.diff(rptr)
    ncol = np.diff(sptr)
    nnzl = sum(int(mm - c + j) for mm, c in zip(m, ncol) for j in range(1, c + 1))
    flops = sum(int(mm - c + j) ** 2 for mm, c in zip(m, ncol) for j in range(1, c + 1))
    assert info["nnz_l"] == nnzl and info["flops"] == flops


def test_user_order_is_honoured_up_to_postorder():
    A = matgen.poisson2d(16)
    geo = matgen.geometric_nd_order((16, 16))
    f, val = make_case(A, nb=16, nemin=4, order=geo)
    assert f.sym_info()["ordering"] == "user"
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


def test_prune_tree_marks():
    """spllt_prune_tree restated (analyse_mod:806-987): ncpu=1 turns every tree
    root with its whole subtree into one pruned subtree (SURVEY 3.1 trap);
    ncpu>1 leaves a top tree with small==0."""
    A = matgen.poisson2d(32)
    f1, _ = make_case(A, nb=32, nemin=16, prune=True, ncpu=1)
    small, sparent = f1.sym("small"), f1.sym("sparent")
    nn = len(sparent)
    roots = [s for s in range(nn) if sparent[s] == nn]
    assert all(small[r] == 1 for r in roots)
    assert all(small[s] == 1 or small[s] < 0 for s in range(nn))
    f4, _ = make_case(A, nb=32, nemin=16, prune=True, ncpu=4)
    small = f4.sym("small")
    assert (small == 0).sum() >= 1 and (small == 1).sum() >= 4
    # members of a subtree point at its root, which is an ancestor
    for s in range(nn):
        if small[s] < 0:
            r = -small[s] - 1
            a = s
            while a != r and a < nn:
                a = f4.sym("sparent")[a] if False else sparent_of(f4, a)
            assert a == r


def sparent_of(f, a, _cache={}):
    key = id(f)
    if key not in _cache:
        _cache[key] = f.sym("sparent")
    return int(_cache[key][a])


def _access_sets(f):
    """Per launch: (reads, writes, atomics) as sets of resources.  A block
    column b is two resources: 2b = its diagonal-tile rows (stored rows <
    width), 2b+1 = the rows below; 2*nbcol + b = the dinv slots of b.  This is
    the granularity at which the stream DAG must order conflicting launches."""
    launches = f.program("launches")
    potrf, units, tiles = f.program("potrf"), f.program("units"), f.program("tiles")
    strips = f.program("strips")
    off, bw = f.sym("bcol_off"), f.sym("bcol_width")
    nbc = len(off)

    def parts(b, r0, cnt):
        out = set()
        if cnt <= 0:
            return out
        if r0 < bw[b]:
            out.add(2 * b)
        if r0 + cnt > bw[b]:
            out.add(2 * b + 1)
        return out

    out = []
    for kind, level, first, count, tile, _fl, st, w0, w1, rec in launches:
        R, W, At = set(), set(), set()
        if kind == 0:
            for q in potrf[first:first + count]:
                b = int(np.searchsorted(off, q["off"], side="right") - 1)
                W.add(2 * b)
                R.add(2 * b)
                W.add(2 * nbc + b)
        elif kind == 4:
            for q in f.program("chains")[first:first + count]:
                b = int(np.searchsorted(off, q["off"], side="right") - 1)
                W |= {2 * b, 2 * nbc + b}
                R.add(2 * b)
        elif kind == 3:
            for uid in sorted(set(tiles[first:first + count]["unit"].tolist())):
                q = strips[uid]
                b = int(np.searchsorted(off, q["off"], side="right") - 1)
                R |= {2 * b, 2 * nbc + b}
                W.add(2 * b + 1)
        elif kind == 1:
            for uid in sorted(set(tiles[first:first + count]["unit"].tolist())):
                u = units[uid]
                db = int(np.searchsorted(off, u["d_off"], side="right") - 1)
                for sg in range(int(u["nseg"])):
                    sb = int(u["src_bcol0"]) + sg
                    sh = int(u["seg_r0"]) + sg * int(u["seg_stride"])
                    R |= parts(sb, int(u["src_r0"]) - sh, int(u["M"]))
                    if u["mode"] != 2:
                        R |= parts(sb, int(u["src_c0"]) - sh, int(u["N"]))
                if u["mode"] == 2:
                    R.add(2 * nbc + db)
                    W |= parts(db, int(u["d_row0"]), int(u["M"]))
                elif u["mode"] == 1:
                    At |= {2 * db, 2 * db + 1}
                elif u["atomic"]:
                    At |= parts(db, int(u["d_row0"]), int(u["M"]))
                else:
                    # plain read-modify-write: the launch must own the destination
                    W |= parts(db, int(u["d_row0"]), int(u["M"]))
        elif kind == 5:
            panels = f.program("panels")
            for uid in sorted(set(tiles[first:first + count]["unit"].tolist())):
                q = panels[uid]
                b = int(np.searchsorted(off, q["off"], side="right") - 1)
                rb, nr = int(q["c0"]) + int(q["pn"]), int(q["nrows"])
                R |= parts(b, rb, nr) | {2 * b, 2 * nbc + b}
                W |= parts(b, rb, nr)
                if q["s_off"] >= 0:
                    sb = int(np.searchsorted(off, q["s_off"], side="right") - 1)
                    R |= parts(sb, rb + int(q["s_rshift"]), nr)
                if q["d_off"] >= 0:
                    db = int(np.searchsorted(off, q["d_off"], side="right") - 1)
                    At |= parts(db, rb - int(q["d_rshift"]), nr)
        out.append((R, W, At))
    return launches, out


@pytest.mark.parametrize("gen,nb,pw", [(lambda: matgen.nd_like((9, 8, 8), 2), 32, 16),
                                        (lambda: matgen.poisson2d(40), 16, 16),
                                        (lambda: matgen.poisson3d(9), 24, 8),
                                        (lambda: matgen.nd_like((9, 8, 8), 2), 8, 8)])  # many block columns per node
@pytest.mark.parametrize("flags", [0, 4, 12, 16, 32])
def test_stream_dag_orders_every_conflict(gen, nb, pw, flags):
    """Two-stream lookahead program: any two launches that touch the same block
    column (write/write, read/write, atomic/plain) must be ordered by stream
    order or an event edge; concurrent atomics into one destination are fine."""
    A = gen()
    f, val = make_case(A, nb=nb, nemin=8, panel_width=pw, engine_flags=flags)
    launches, acc = _access_sets(f)
    n = len(launches)
    assert (launches[:, 6] == 1).any(), "expected bulk-stream launches in this case"
    assert ((launches[:, 0] == 3).any()) == (flags in (4, 12)), "fused strip launches only with flag 4"
    assert ((launches[:, 0] == 4).any()) == (flags == 4 and pw % 16 == 0), \
        "tile-chain launches with flag 4 (bit 3 disables; panel widths that are no multiple of 16 too)"
    assert ((launches[:, 0] == 5).any()) == (flags == 32), "fused panel steps with bit 5 (not in strip mode)"
    rec_at = {}
    last_in_stream = {}
    before = [0] * n  # bitset of launches that happen-before launch i
    for i, (kind, level, first, count, tile, _fl, st, w0, w1, rec) in enumerate(launches):
        m = 0
        if st in last_in_stream:
            j = last_in_stream[st]
            m |= before[j] | (1 << j)
        for w in (w0, w1):
            if w >= 0:
                j = rec_at[w]          # a wait must refer to an earlier record
                m |= before[j] | (1 << j)
        before[i] = m
        last_in_stream[st] = i
        if rec >= 0:
            rec_at[int(rec)] = i
    for j in range(n):
        Rj, Wj, Aj = acc[j]
        for i in range(j):
            Ri, Wi, Ai = acc[i]
            conflict = (Wi & (Rj | Wj | Aj)) or (Wj & (Ri | Ai)) or (Ai & Rj) or (Aj & Ri)
            if conflict:
                assert before[j] >> i & 1, (i, j, launches[i].tolist(), launches[j].tolist())
    # the final event covers everything: last launch of each stream precedes it
    fin = max(rec_at.values())
    for st, i in last_in_stream.items():
        assert i == fin or (before[fin] >> i & 1) or st == launches[fin, 6]


def test_single_stream_program_has_no_events():
    A = matgen.poisson2d(20)
    f, val = make_case(A, nb=8, nemin=4, engine_flags=2)
    L = f.program("launches")
    assert (L[:, 6] == 0).all() and (L[:, 7:] == -1).all()
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


@pytest.mark.parametrize("flags", [0, 4, 2, 6, 12, 16, 32, 34])
def test_program_variants_agree(flags):
    """fused strip / per-panel TRSM, two-stream / single-stream programs all
    reproduce the same factor (interpreted in numpy)."""
    A = matgen.nd_like((8, 7, 7), 2)
    f, val = make_case(A, nb=48, nemin=8, panel_width=16, engine_flags=flags)
    L = f.program("launches")
    assert ((L[:, 0] == 3).any()) == (flags in (4, 12))   # strip kernel needs the two-stream program
    # fused panel steps with bit 5 unless the strip mode (bit 2, two-stream only) is active
    assert ((L[:, 0] == 5).any()) == (bool(flags & 32) and not ((flags & 4) and not (flags & 2)))
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


@pytest.mark.parametrize("flags", [0, 64])
def test_inter_node_updates_are_sliced_over_the_far_stream(flags):
    """Default program: the inter-node update of a node is issued in K slices on
    stream 2 while the level's panel chains still run (nodes that finish early
    go completely); engine flag 64 keeps one launch per level.  Both reproduce
    the oracle's factor when interpreted in numpy."""
    A = matgen.nd_like((9, 8, 8), 2)
    f, val = make_case(A, nb=8, nemin=8, panel_width=8, engine_flags=flags)
    L = f.program("launches")
    far = L[L[:, 6] == 2]
    if flags == 0:
        assert len(far) >= 4
        units, tiles = f.program("units"), f.program("tiles")
        nseg = [int(units[int(tiles[int(l[2])]["unit"])]["nseg"]) for l in far if l[3] > 0]
        assert min(nseg) >= 1 and max(nseg) >= 2          # K slices of several block columns
        assert all(units[int(tiles[int(l[2])]["unit"])]["mode"] == 1 for l in far if l[3] > 0)
    else:
        assert len(far) == 0
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13
