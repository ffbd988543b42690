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
    """Per launch: (reads, writes, atomics), each a list of resources.  A resource is a
    rectangle (block column, stored rows [r0, r1), columns [c0, c1)) of the arena or a Winv
    slot ("w", block column, panel).  This is the granularity at which the stream DAG must
    order conflicting launches: chain steps, side launches and the updates of one block
    column work on disjoint rectangles of it at the same time."""
    launches = f.program("launches")
    units, tiles, chains = f.program("units"), f.program("tiles"), f.program("chains")
    off, bw, bnr = f.sym("bcol_off"), f.sym("bcol_width"), f.sym("bcol_nrow")
    pw = int(f.program("panel_width"))

    def bcol_of(o):
        return int(np.searchsorted(off, o, side="right") - 1)

    out = []
    for kind, level, first, count, tile in launches[:, :5]:
        R, W, At = [], [], []
        if kind == 4:
            for q in chains[first:first + count]:
                b = bcol_of(q["off"])
                c0, pn, cs, ce = int(q["c0"]), int(q["pn"]), int(q["cs"]), int(q["ce"])
                R.append((b, c0, c0 + pn, cs, c0))
                R.append((b, c0, ce, c0, ce))
                W.append((b, c0, ce, c0, ce))
                W.append(("wi", b, c0 // pw))      # inverse part of Winv
        elif kind == 8:
            # a chain block of up to four panels: factor of its diagonal block + the panels' inverses
            for q in chains[first:first + count]:
                b = bcol_of(q["off"])
                c0, cw = int(q["c0"]), int(q["pn"])
                R.append((b, c0, c0 + cw, c0, c0 + cw))
                W.append((b, c0, c0 + cw, c0, c0 + cw))
                for p in range(c0 // pw, (c0 + cw - 1) // pw + 1):
                    W.append(("wi", b, p))
        elif kind == 10:
            # subtree tasks: a task owns the block columns of its nodes (read + written by it alone) and
            # its generated element; what its root sends up lands in the ancestors with atomics
            snodes = f.program("sub_nodes")
            for t in f.program("sub_tasks")[first:first + count]:
                W.append(("gen", int(t["g_off"]), 0))
                for nd in snodes[int(t["node_first"]):int(t["node_first"]) + int(t["node_count"])]:
                    b = bcol_of(nd["off"])
                    assert int(bw[b]) == nd["w"] and int(bnr[b]) == nd["nrow"]
                    W.append((b, 0, int(bnr[b]), 0, int(bw[b])))
                    W.append(("wi", b, 0))
                    for u in units[int(nd["unit_first"]):int(nd["unit_first"]) + int(nd["unit_count"])]:
                        if u["mode"] == 1 and nd["root"]:
                            db = bcol_of(u["d_off"])
                            At.append((db, 0, int(bnr[db]), 0, int(bw[db])))
        elif kind == 2:
            # exchange: the pack reads, the unpack overwrites whole block columns (engine.cpp
            # pre_exchange / post_exchange)
            xk, xfirst, xn = (int(v) for v in f.program("exchanges")[first][:3])
            rank = getattr(f, "rank", 0)
            for b, root, xo, cnt, xoff_, space in f.program("xitems")[xfirst:xfirst + xn].tolist():
                if space:      # the block column's dinv slots: the inverses of all its panels
                    whole = [("wi", b, p) for p in range(-(-int(bw[b]) // pw))]
                else:
                    assert cnt == int(bnr[b]) * int(bw[b]) and xoff_ == off[b]
                    whole = [(b, 0, int(bnr[b]), 0, int(bw[b]))]
                if xk in (0, 1) or (xk == 2 and root == rank):
                    R.extend(whole)
                if xk == 0 or (xk == 1 and root == rank) or (xk == 2 and root != rank):
                    W.extend(whole)
        elif kind == 7:
            pu = f.program("panels")
            tl = tiles[first:first + count]
            for uid in sorted(set(tl["unit"].tolist())):
                q = pu[uid]
                b = bcol_of(q["off"])
                c0, pn, pn2, nrow = int(q["c0"]), int(q["pn"]), int(q["next_pn"]), int(q["nrow"])
                r1 = c0 + pn
                assert int(bw[b]) == q["ld"] and int(bnr[b]) == nrow
                # every 64-row block below the panel has its workgroup (+ one when there is none)
                tis = sorted(tl["ti"][(tl["unit"] == uid) & (tl["tj"] == 0)].tolist())
                assert tis == list(range(max(1, -(-(nrow - r1) // 64)))), (tis, nrow, r1)
                assert len(tis) == q["ntile"]       # the "last reader" counters count to this
                R.append((b, c0, nrow, c0, r1))
                W.append((b, c0, nrow, c0, r1))
                W.append(("wi", b, c0 // pw))
                if pn2 > 0:
                    R.append((b, r1, nrow, 0, c0))
                    R.append((b, r1, nrow, r1, r1 + pn2))
                    W.append((b, r1, nrow, r1, r1 + pn2))
        elif kind == 6:
            gt = f.program("gather_tiles")
            R.append(("scratch", 0, 0))
            for t in gt[first:first + count]:
                b = bcol_of(t["d_off"])
                r0, c0 = int(t["row0"]), int(t["col0"])
                W.append((b, r0, r0 + int(t["rows"]), c0, c0 + int(t["cols"])))
        elif kind in (1, 9):
            for uid in sorted(set(tiles[first:first + count]["unit"].tolist())):
                u = units[uid]
                if kind == 9:
                    # k_trsm_rows also reads the factored diagonal block of its chain block, and every
                    # 64-row block of the unit has its workgroup
                    b9, cs9, cw9 = bcol_of(u["d_off"]), int(u["d_col0"]), int(u["N"])
                    R.append((b9, cs9, cs9 + cw9, cs9, cs9 + cw9))
                    tl9 = tiles[first:first + count]
                    assert sorted(tl9["ti"][tl9["unit"] == uid].tolist()) == list(range(-(-int(u["M"]) // 64)))
                db = bcol_of(u["d_off"]) if u["mode"] != 3 else -1
                M, N = int(u["M"]), int(u["N"])
                for sg in range(int(u["nseg"])):
                    sb = int(u["src_bcol0"]) + sg
                    sh = int(u["seg_r0"]) + sg * int(u["seg_stride"])
                    k0 = int(u["k0"]) if u["nseg"] == 1 else 0
                    k1 = k0 + int(u["klen"]) if (u["nseg"] == 1 and u["klen"] >= 0) else int(bw[sb])
                    ra = int(u["src_r0"]) - sh
                    R.append((sb, ra, ra + M, k0, k1))
                    if u["mode"] != 2:
                        rb = int(u["src_c0"]) - sh
                        R.append((sb, rb, rb + N, k0, k1))
                dr0, dc0 = int(u["d_row0"]), int(u["d_col0"])
                if u["mode"] == 3:
                    W.append(("scratch", 0, 0))
                    continue
                if u["mode"] == 2:
                    for p in range(dc0 // pw, (dc0 + N - 1) // pw + 1):    # the inverses of the panels it solves with
                        R.append(("wi", db, p))
                    if int(u["klen"]) > N:
                        R.append(("ww", db, dc0 // pw))
                    W.append((db, dr0, dr0 + M, dc0, dc0 + N))
                elif u["mode"] == 1:
                    At.append((db, 0, int(bnr[db]), 0, int(bw[db])))
                elif u["atomic"]:
                    At.append((db, dr0, dr0 + M, dc0, dc0 + N))
                else:
                    # plain read-modify-write: the launch must own the destination
                    W.append((db, dr0, dr0 + M, dc0, dc0 + N))
        out.append((R, W, At))
    return launches, out


def _overlap(a, b):
    if isinstance(a[0], str) or isinstance(b[0], str):
        return a == b
    return (a[0] == b[0] and a[1] < b[2] and b[1] < a[2] and a[3] < b[4] and b[3] < a[4] and
            a[1] < a[2] and a[3] < a[4] and b[1] < b[2] and b[3] < b[4])


def _any_overlap(xs, ys):
    if not xs or not ys:
        return False
    by = {}
    for y in ys:
        by.setdefault(y[1] if isinstance(y[0], str) else y[0], []).append(y)
    return any(_overlap(x, y) for x in xs for y in by.get(x[1] if isinstance(x[0], str) else x[0], ()))


def dag_violations(f):
    """Pairs of launches that conflict (write/write, read/write, atomic/plain) without being
    ordered by stream order or an event edge, plus checks of the event bookkeeping."""
    launches, acc = _access_sets(f)
    n = len(launches)
    rec_at, last_in_stream = {}, {}
    before = [0] * n  # bitset of launches that happen-before launch i
    for i, l in enumerate(launches):
        st, rec, waits = int(l[6]), int(l[7]), l[8:12]
        m = 0
        if st in last_in_stream:
            j = last_in_stream[st]
            m |= before[j] | (1 << j)
        for w in waits:
            if w >= 0:
                j = rec_at[int(w)]          # a wait must refer to an earlier record
                m |= before[j] | (1 << j)
        before[i] = m
        last_in_stream[st] = i
        if rec >= 0:
            assert rec not in rec_at, "every event is recorded once"
            rec_at[rec] = i
    bad = []
    for j in range(n):
        Rj, Wj, Aj = acc[j]
        for i in range(j):
            if before[j] >> i & 1:
                continue
            Ri, Wi, Ai = acc[i]
            if (_any_overlap(Wi, Rj) or _any_overlap(Wi, Wj) or _any_overlap(Wi, Aj) or
                    _any_overlap(Wj, Ri) or _any_overlap(Wj, Ai) or _any_overlap(Ai, Rj) or
                    _any_overlap(Aj, Ri)):
                bad.append((i, j, launches[i].tolist(), launches[j].tolist()))
    return bad, launches, before, rec_at, last_in_stream


_DAG_GENS = {"nd-32-16": (lambda: matgen.nd_like((9, 8, 8), 2), 32, 16),
             "p2d-16-16": (lambda: matgen.poisson2d(40), 16, 16),
             "p3d-24-8": (lambda: matgen.poisson3d(9), 24, 8),
             "nd-8-8": (lambda: matgen.nd_like((9, 8, 8), 2), 8, 8),      # many block columns per node
             "nd-100-8": (lambda: matgen.nd_like((9, 8, 8), 2), 100, 8)}
_DAG_FLAGS = [0, 64, 512, 1024, 2048, 2560, 4096, 4160, 4608]


def _dag_cases():
    """every engine variant x both chain layouts on two structures, the main variants on the other
    three, the (ignored) chain block knob on one: the full cross product (360 cases, most of an hour
    of Python rectangle checks) found nothing the subset does not"""
    out = []
    for g in ("nd-32-16", "nd-8-8"):
        out += [(g, 0, fl, c2) for fl in _DAG_FLAGS for c2 in (1, 0)]
    for g in ("p2d-16-16", "p3d-24-8", "nd-100-8"):
        out += [(g, 0, fl, 0) for fl in (0, 512, 4096)] + [(g, 0, 0, 1)]
    out += [("nd-32-16", cb, 0, c2) for cb in (8, 40) for c2 in (1, 0)]
    return out


@pytest.mark.parametrize("case,cb,flags,chain4", _dag_cases())
def test_stream_dag_orders_every_conflict(case, cb, flags, chain4, monkeypatch):
    gen, nb, pw = _DAG_GENS[case]
    _stream_dag_orders_every_conflict(gen, nb, pw, cb, flags, chain4, monkeypatch)


def _stream_dag_orders_every_conflict(gen, nb, pw, cb, flags, chain4, monkeypatch):
    """Multi-stream program (chain, side, bulk, far, wide): any two launches that touch the
    same entries (write/write, read/write, atomic/plain) must be ordered by stream order or
    an event edge; concurrent atomics into one destination are fine.  cb: the (ignored) chain
    block knob; chain4: chain blocks of up to four panels (k_chain_block + k_trsm_rows; the default)
    for steps with block columns of several panels, or one panel per chain step throughout."""
    if cb:
        monkeypatch.setenv("SPLLT_CHAIN_BLOCK", str(cb))
    monkeypatch.setenv("SPLLT_CHAIN4", str(chain4))
    A = gen()
    f, val = make_case(A, nb=nb, nemin=8, panel_width=pw, engine_flags=flags)
    bad, launches, before, rec_at, last_in_stream = dag_violations(f)
    assert not bad, bad[:3]
    assert (launches[:, 6] == 1).any(), "expected bulk-stream launches in this case"
    assert ((launches[:, 6] == 3) == (launches[:, 0] == 10)).all(), "the side stream carries the subtree tasks only"
    kinds = launches[:, 0]
    if chain4 and nb > pw:       # block columns of several panels: chain blocks + their row solves
        assert (kinds == 8).any() and (kinds == 9).any()
    else:
        assert not (kinds == 8).any() and not (kinds == 9).any()
    if not chain4:
        fused = not flags & 512      # fused panel launches replace the chain steps (the chain block knob is ignored)
        assert (kinds == (7 if fused else 4)).any() and not (kinds == (4 if fused else 7)).any()
    assert ((launches[:, 0] == 6).any()) == bool(flags & 4096), "gather launches only in the deterministic engine"
    if flags & 4096:
        units = f.program("units")
        assert not (units["mode"] == 1).any() and not units["atomic"].any(), "no atomic unit at all"
    # the final event covers everything: last launch of each stream precedes it
    fin = max(rec_at.values())
    for st, i in last_in_stream.items():
        assert i == fin or (before[fin] >> i & 1) or st == launches[fin, 6]
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


def test_single_stream_program_has_no_events():
    A = matgen.poisson2d(20)
    f, val = make_case(A, nb=8, nemin=4, engine_flags=2)
    L = f.program("launches")
    assert (L[:, 6] == 0).all() and (L[:, 7:] == -1).all()   # one stream, no record, no waits
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


@pytest.mark.parametrize("flags", [0, 2, 64, 66, 512, 514, 1024, 2048, 4096, 4098, 4608])
@pytest.mark.parametrize("cb", [0, 16, 32])
@pytest.mark.parametrize("chain4", [1, 0])
def test_program_variants_agree(flags, cb, chain4, monkeypatch):
    """multi-stream / single-stream programs, with and without early inter-node slices, zone
    pipeline forced on / off, deterministic engine, chain blocks of up to four panels or one panel
    per chain step: all reproduce the same factor (interpreted in numpy)."""
    if cb:
        monkeypatch.setenv("SPLLT_CHAIN_BLOCK", str(cb))
    monkeypatch.setenv("SPLLT_CHAIN4", str(chain4))
    A = matgen.nd_like((8, 7, 7), 2)
    f, val = make_case(A, nb=48, nemin=8, panel_width=16, engine_flags=flags)
    assert f.program("chain_block") == 16 and f.program("panel_width") == 16   # the inverses stay per panel
    kinds = f.program("launches")[:, 0]
    assert not (kinds == 5).any()
    if chain4:
        assert (kinds == 8).any() and (kinds == 9).any()      # (48-wide block columns: three panels)
    else:
        assert not (kinds == 8).any() and (kinds == 7).any() == (not flags & 512), "fused panel launches unless flag 512"
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


@pytest.mark.parametrize("flags", [0, 64])
def test_inter_node_updates_are_sliced_over_the_far_stream(flags):
    """The inter-node updates run on the far stream.  At the end of a level they are issued
    sorted by destination zone, one launch + event per zone (block column c of the nodes of
    the next level), so that step c of the next level only waits for zone c.  Default program:
    part of a node's update is issued even earlier, in K slices, while the level's panel
    chains still run (nodes that finish early go completely); engine flag 64 turns the slices
    off.  Both reproduce the oracle's factor when interpreted in numpy."""
    A = matgen.nd_like((9, 8, 8), 2)
    f, val = make_case(A, nb=8, nemin=8, panel_width=8, engine_flags=flags)
    L = f.program("launches")
    far = L[(L[:, 6] == 2) & (L[:, 3] > 0)]
    units, tiles = f.program("units"), f.program("tiles")
    node_bc0, bc_node = f.sym("node_bcol0"), f.sym("bcol_node")
    assert len(far) >= 4
    partial = 0
    for l in far:
        for uid in sorted(set(tiles[int(l[2]):int(l[2]) + int(l[3])]["unit"].tolist())):
            u = units[uid]
            assert u["mode"] == 1
            s = int(bc_node[int(u["src_bcol0"])])
            if int(u["nseg"]) < int(node_bc0[s + 1] - node_bc0[s]):
                partial += 1
    assert (partial > 0) == (flags == 0)     # K slices only in the default program
    # one event per zone launch, and the chain steps of later levels wait for far-stream events
    far_events = set(int(e) for e in L[L[:, 6] == 2][:, 7] if e >= 0)
    chain_waits = set(int(w) for w in L[np.isin(L[:, 0], (4, 7, 8))][:, 8:12].ravel() if w >= 0)   # chain / fused panel / chain-block steps
    assert len(far_events & chain_waits) >= 3
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


_SUB_CASES = {"p2d-40": (lambda: matgen.poisson2d(40), 64, 64, 8),        # deep subtrees of tiny nodes
              "p2d-12": (lambda: matgen.poisson2d(12), 64, 64, 4),        # the whole tree fits one task
              "nd-32-16": (lambda: matgen.nd_like((9, 8, 8), 2), 32, 16, 8),
              "p3d-24-8": (lambda: matgen.poisson3d(9), 24, 8, 8)}


@pytest.mark.parametrize("flags", [0, 2, 64, 512])
@pytest.mark.parametrize("budget", [40, 300, 100000])
@pytest.mark.parametrize("case", sorted(_SUB_CASES))
def test_subtree_tasks_program(case, budget, flags, monkeypatch):
    """Small subtrees as single device tasks (L_SUBTREE, the reference's a20-a25 shape; opt-in): every
    node belongs to a task or to the level-batched program, never both; a task holds a whole subtree
    (children before parents, the root last); what leaves the subtree from below the root goes through
    the task's generated element (MODE_GEN) and the root takes it to the ancestors; the stream DAG
    orders the tasks (side stream) against everything that touches their block columns or their
    ancestors non-atomically; the interpreted program reproduces the dense factor."""
    monkeypatch.setenv("SPLLT_SUBTREES", "1")
    monkeypatch.setenv("SPLLT_SUBTREE_US", str(budget))
    gen, nb, pw, nemin = _SUB_CASES[case]
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw, engine_flags=flags)
    L = f.program("launches")
    tasks, snodes, units = f.program("sub_tasks"), f.program("sub_nodes"), f.program("units")
    assert (L[:, 0] == 10).sum() == 1 and L[0, 0] == 10 and L[0, 3] == len(tasks) > 0
    assert L[0, 6] == (0 if flags & 2 else 3)            # side stream of the multi-stream program
    off, sparent = f.sym("bcol_off"), f.sym("sparent")
    node_bc0, bc_node = f.sym("node_bcol0"), f.sym("bcol_node")
    in_task = {}
    for ti, t in enumerate(tasks):
        nd = snodes[int(t["node_first"]):int(t["node_first"]) + int(t["node_count"])]
        ids = [int(bc_node[int(np.searchsorted(off, q["off"]))]) for q in nd]
        assert ids == sorted(ids) and nd[-1]["root"] == 1 and not nd[:-1]["root"].any()
        for s in ids:
            assert s not in in_task and node_bc0[s + 1] - node_bc0[s] == 1
            in_task[s] = ti
        root = ids[-1]
        for s in ids[:-1]:                                 # a whole subtree: every parent chain ends in the root
            a = s
            while a != root:
                a = int(sparent[a])
                assert a in ids
        kids = [c for c in range(len(sparent)) if int(sparent[c]) in ids]
        assert all(c in ids for c in kids), "a task holds all descendants of its root"
        gen_units = 0
        for q in nd:
            us = units[int(q["unit_first"]):int(q["unit_first"]) + int(q["unit_count"])]
            gen_units += int((us["mode"] == 4).sum())
            assert not (q["root"] and (us["mode"] == 4).any())
        below_root = int(nd[-1]["nrow"]) - int(nd[-1]["w"])      # (0: the root of a tree of the forest)
        assert t["g_n"] == (below_root if len(ids) > 1 else 0) and (gen_units > 0) <= (t["g_n"] > 0)
    # nothing of a task's nodes appears in the level-batched part
    tiles = f.program("tiles")
    for kind, level, first, count, tile in L[1:, :5]:
        if kind == 1 and count > 0:
            for uid in set(tiles[first:first + count]["unit"].tolist()):
                assert int(bc_node[int(units[uid]["src_bcol0"])]) not in in_task
    if budget == 100000 and case == "p2d-12":
        assert len(in_task) == f.sym_info()["nnodes"], "the whole forest in tasks: the program ends behind them"
    if not flags & 2:
        bad, launches, before, rec_at, last_in_stream = dag_violations(f)
        assert not bad, bad[:3]
        fin = max(rec_at.values())
        for st, i in last_in_stream.items():
            assert i == fin or (before[fin] >> i & 1) or st == launches[fin, 6]
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13


def test_subtree_tasks_are_opt_in(monkeypatch):
    monkeypatch.delenv("SPLLT_SUBTREES", raising=False)
    f, val = make_case(matgen.poisson2d(40), nb=64, nemin=8)
    assert not (f.program("launches")[:, 0] == 10).any() and f.program("gen_size") == 0
    f2, _ = make_case(matgen.poisson2d(40), nb=64, nemin=8, engine_flags=262144)     # bit 18
    assert (f2.program("launches")[:, 0] == 10).any() and f2.program("gen_size") > 0
    for fl in (4096, 262144 | 4096):                                                  # never in the deterministic engine
        f3, _ = make_case(matgen.poisson2d(40), nb=64, nemin=8, engine_flags=fl)
        assert not (f3.program("launches")[:, 0] == 10).any()


def test_chain_replay_gets_the_single_stream_program(monkeypatch):
    """A factorization small enough to be replayed as ONE chain of graph nodes (the default up to 40 GFLOP)
    is built as the single-stream program -- no zones, slices, markers or events: fewer kernels in the
    chain -- unless the caller asks for eager launches / the DAG replay, or the problem is large."""
    monkeypatch.setenv("SPLLT_CHAIN_GRAPH_SERIAL", "1")
    A = matgen.poisson2d(40)
    f, val = make_case(A, nb=64, nemin=8)
    L = f.program("launches")
    assert (L[:, 6] == 0).all() and (L[:, 7:] == -1).all() and (L[:, 3] > 0).all()
    f2, _ = make_case(A, nb=64, nemin=8, engine_flags=2)                  # the explicit single-stream program
    assert np.array_equal(L, f2.program("launches"))
    for fl in (65536, 131072):                                               # DAG replay / eager: multi-stream
        f3, _ = make_case(A, nb=64, nemin=8, engine_flags=fl)
        L3 = f3.program("launches")
        assert (L3[:, 6] != 0).any() and (L3[:, 7] >= 0).any() and len(L3) > len(L)
    monkeypatch.setenv("SPLLT_CHAIN_GRAPH_SERIAL", "0")
    f4, _ = make_case(A, nb=64, nemin=8)
    assert len(f4.program("launches")) > len(L)
    got = emulate_program(f, val)
    assert rel_err(got, dense_arena(f, A), lower_mask(f)) < 1e-13
