"""Numpy interpreter of the stream-DAG program exported by
spllt_hip_program_get.  TEST-ONLY: it executes the same work tables the HIP
kernels consume (potrf units, update units, tiles, launches) with dense numpy
arithmetic, so that the host-side scheduler can be validated on a machine
without a GPU.  It is not part of the product and is never timed."""
import numpy as np
import scipy.linalg as sl

MODE_DIRECT, MODE_SCATTER, MODE_TRSM = 0, 1, 2


def emulate_program(f, val, exchange=None, partitioned=False):
    """exchange(k, xbuf) -> xbuf after the collective is called at every EXCHANGE launch of a
    partitioned (multi-GPU) program: k = index into f.program("exchanges"), xbuf = this rank's
    packed exchange buffer (spllt_amd.multigpu.run_exchange is the production collective)."""
    info = f.sym_info()
    arena = np.zeros(info["arena"])
    md, ms = f.sym("map_dst"), f.sym("map_src")
    if partitioned:
        keep = f.partition("map_keep").astype(bool)
        md, ms = md[keep], ms[keep]
    arena[md] = np.asarray(val)[ms]
    launches = f.program("launches")
    potrf = f.program("potrf")
    units = f.program("units")
    tiles = f.program("tiles")
    relpos = f.program("relpos")
    rlist = f.sym("rlist")
    bc_off, bc_w = f.sym("bcol_off"), f.sym("bcol_width")
    dinv = np.zeros(max(1, f.program("dinv_size")))
    scratch = np.zeros(max(1, f.program("scratch_size")))   # MODE_BUFFER products (deterministic engine)

    def seg_rows(u, sg, r0, cnt, from_b):
        """rows [r0, r0+cnt) x K-window of segment sg as a dense (cnt x klen) array"""
        if from_b and u["b_bcol0"] >= 0:
            bcol = int(u["b_bcol0"]) + sg
            rshift = int(u["b_seg_r0"]) + sg * int(u["seg_stride"])
        else:
            bcol = int(u["src_bcol0"]) + sg
            rshift = int(u["seg_r0"]) + sg * int(u["seg_stride"])
        w = int(bc_w[bcol])
        kbeg = int(u["k0"]) if u["nseg"] == 1 else 0
        klen = int(u["klen"]) if (u["nseg"] == 1 and u["klen"] >= 0) else w
        base = int(bc_off[bcol])
        out = np.empty((cnt, klen))
        for i in range(cnt):
            s = base + (r0 + i - rshift) * w + kbeg
            out[i] = arena[s:s + klen]
        return out

    bc_nrow = f.sym("bcol_nrow")
    gen = None                                               # generated elements of the subtree tasks
    for kind, level, first, count, tile in launches[:, :5]:
        if kind == 2:  # EXCHANGE k: pack, the caller's collective, unpack (engine.cpp pre_/post_exchange)
            xk, xfirst, xn, xelems, xchunk = (int(v) for v in f.program("exchanges")[first])
            items = f.program("xitems")[xfirst:xfirst + xn]
            rank = getattr(f, "rank", 0)
            xbuf = np.zeros(max(1, f.program("xbuf_elems")))
            for b, root, xo, cnt, off, space in items:
                if xk == 2 and root != rank:
                    continue                      # the owner sends
                xbuf[xo:xo + cnt] = (dinv if space else arena)[off:off + cnt]
            # kinds 0 and 3: the last element is the "not positive definite" indicator (0 = fine)
            xbuf = exchange(first, xbuf)
            for b, root, xo, cnt, off, space in items:
                if (xk == 2 and root == rank) or (xk == 1 and root != rank):
                    continue
                (dinv if space else arena)[off:off + cnt] = xbuf[xo:xo + cnt]
            continue
        if kind == 10:  # one small subtree per workgroup (k_subtree): nodes in post-order
            tasks, snodes = f.program("sub_tasks"), f.program("sub_nodes")
            if gen is None:
                gen = np.zeros(max(1, f.program("gen_size")))
            for t in tasks[first:first + count]:
                for nd in snodes[int(t["node_first"]):int(t["node_first"]) + int(t["node_count"])]:
                    w, m, off = int(nd["w"]), int(nd["nrow"]), int(nd["off"])
                    blk = arena[off:off + m * w].reshape(m, w)          # (view: a node has ONE block column)
                    dd = np.tril(blk[:w])
                    Lb = sl.cholesky(dd + np.tril(dd, -1).T, lower=True)
                    blk[:w][np.tril_indices(w)] = Lb[np.tril_indices(w)]
                    inv = sl.solve_triangular(Lb, np.eye(w), lower=True)
                    do = int(nd["dinv_off"])
                    dinv[do:do + w * w] = inv.ravel()
                    blk[w:] = blk[w:] @ inv.T
                    for u in units[int(nd["unit_first"]):int(nd["unit_first"]) + int(nd["unit_count"])]:
                        assert u["nseg"] == 1 and u["klen"] < 0 and u["lower"] and u["src_r0"] == u["src_c0"]
                        r0, M, N = int(u["src_r0"]), int(u["M"]), int(u["N"])
                        P = blk[r0:r0 + M] @ blk[r0:r0 + N].T
                        ii, jj = np.arange(M)[:, None], np.arange(N)[None, :]
                        keep = ii >= jj
                        if u["mode"] == 4:       # MODE_GEN: into the subtree's generated element
                            assert not nd["root"] and u["d_off"] == t["g_off"] and u["relrow_off"] == u["gcol_off"]
                            gr = relpos[int(u["relrow_off"]) + ii]
                            gc = relpos[int(u["gcol_off"]) + jj]
                            assert (gr < t["g_n"]).all() and (gr >= 0).all() and (np.diff(gr[:, 0]) > 0).all()
                            idx = int(u["d_off"]) + gr * (gr + 1) // 2 + gc
                            np.subtract.at(gen, idx[keep], P[keep])
                            continue
                        assert u["mode"] == MODE_SCATTER
                        dr = relpos[int(u["relrow_off"]) + ii] - int(u["d_row0"])
                        dc = rlist[int(u["gcol_off"]) + jj] - int(u["d_col0"])
                        assert (dr >= 0).all() and (dc >= 0).all() and (dc < u["d_ld"]).all()
                        if nd["root"] and t["g_n"] > 0:
                            gi, gj = r0 - w + ii, r0 - w + jj
                            gidx = int(t["g_off"]) + gi * (gi + 1) // 2 + gj
                            P = P.copy()
                            P[keep] -= gen[gidx[keep]]
                            gen[gidx[keep]] = 0.0
                        idx = int(u["d_off"]) + dr * int(u["d_ld"]) + dc
                        np.subtract.at(arena, idx[keep], P[keep])
            assert not gen.any(), "the generated elements are zero again when the launch ends"
            continue
        if kind == 4:  # one step of the panel chain per unit (k_chain_panel)
            for q in f.program("chains")[first:first + count]:
                ld, off = int(q["ld"]), int(q["off"])
                c0, pn, cs, ce = int(q["c0"]), int(q["pn"]), int(q["cs"]), int(q["ce"])
                cq = c0 - cs

                def idx(r0, r1, k0, k1):
                    return off + np.arange(r0, r1)[:, None] * ld + np.arange(k0, k1)[None, :]
                dd = idx(c0, c0 + pn, c0, c0 + pn)
                blk = np.tril(arena[dd])
                Lb = sl.cholesky(blk + np.tril(blk, -1).T, lower=True)
                low = np.tril_indices(pn)
                arena[dd[low]] = Lb[low]
                inv = sl.solve_triangular(Lb, np.eye(pn), lower=True)
                wo = int(q["winv_off"])
                ldw = ce - cs                                            # row stride: the chain block's width
                Wv = dinv[wo:wo + pn * ldw].reshape(pn, ldw)             # view: the panel's rows of the block's inverse
                Wv[:, cq:cq + pn] = inv
                if ce > c0 + pn:
                    xi = idx(c0 + pn, ce, c0, c0 + pn)
                    X = arena[xi] @ inv.T
                    arena[xi] = X
                    ti = idx(c0 + pn, ce, c0 + pn, ce)
                    keep = np.tril(np.ones((ce - c0 - pn,) * 2, dtype=bool))
                    arena[ti[keep]] -= (X @ X.T)[keep]
            continue
        if kind == 8:  # a chain block of up to four panels per unit (k_chain_block): factor + per-panel inverses
            pw = f.program("panel_width")
            for q in f.program("chains")[first:first + count]:
                ld, off = int(q["ld"]), int(q["off"])
                c0, cw = int(q["c0"]), int(q["pn"])
                assert int(q["cs"]) == c0 and int(q["ce"]) == c0 + cw and cw <= 4 * pw and c0 % pw == 0
                dd = off + np.arange(c0, c0 + cw)[:, None] * ld + np.arange(c0, c0 + cw)[None, :]
                blk = np.tril(arena[dd])
                Lb = sl.cholesky(blk + np.tril(blk, -1).T, lower=True)
                low = np.tril_indices(cw)
                arena[dd[low]] = Lb[low]
                wo = int(q["winv_off"])
                for p0 in range(0, cw, pw):
                    n = min(pw, cw - p0)
                    inv = sl.solve_triangular(Lb[p0:p0 + n, p0:p0 + n], np.eye(n), lower=True)
                    dinv[wo:wo + n * n] = inv.ravel()
                    wo += n * n
            continue
        if kind == 7:  # one whole panel step per launch (k_panel), workgroup by workgroup
            pu = f.program("panels")
            snap = arena.copy()      # every workgroup reads the state before the launch
            for t in tiles[first:first + count]:
                q = pu[int(t["unit"])]
                ld, off, nrow = int(q["ld"]), int(q["off"]), int(q["nrow"])
                c0, pn, pn2 = int(q["c0"]), int(q["pn"]), int(q["next_pn"])

                def idx(r0, r1, k0, k1):
                    return off + np.arange(r0, r1)[:, None] * ld + np.arange(k0, k1)[None, :]
                dd = idx(c0, c0 + pn, c0, c0 + pn)
                factored = bool(int(q["pad_"]) & 1)      # a chain launch factored the panel: only its inverse is read
                if factored:
                    do = int(q["dinv_off"])
                    inv = dinv[do:do + pn * pn].reshape(pn, pn).copy()
                else:
                    blk = np.tril(snap[dd])
                    Lb = sl.cholesky(blk + np.tril(blk, -1).T, lower=True)
                    inv = sl.solve_triangular(Lb, np.eye(pn), lower=True)
                ti = int(t["ti"])
                if ti == 0 and not factored:
                    low = np.tril_indices(pn)
                    arena[dd[low]] = Lb[low]
                    do = int(q["dinv_off"])
                    dinv[do:do + pn * pn] = inv.ravel()
                r1 = c0 + pn
                r0 = r1 + 64 * ti
                nr = min(64, nrow - r0)
                if nr <= 0:
                    continue
                Xi = snap[idx(r0, r0 + nr, c0, r1)] @ inv.T
                arena[idx(r0, r0 + nr, c0, r1)] = Xi
                if pn2 <= 0:
                    continue
                Xd = snap[idx(r1, r1 + pn2, c0, r1)] @ inv.T
                upd = (np.hstack([snap[idx(r0, r0 + nr, 0, c0)], Xi]) @
                       np.hstack([snap[idx(r1, r1 + pn2, 0, c0)], Xd]).T)
                ci = idx(r0, r0 + nr, r1, r1 + pn2)
                keep = (r0 + np.arange(nr))[:, None] >= (r1 + np.arange(pn2))[None, :]
                arena[ci[keep]] = snap[ci[keep]] - upd[keep]
            continue
        if kind == 6:  # ordered gather of buffered update blocks into destination tiles (k_gather)
            gt, gi = f.program("gather_tiles"), f.program("gather_items")
            for t in gt[first:first + count]:
                acc = np.zeros((64, 64))
                for g in gi[int(t["first"]):int(t["first"]) + int(t["count"])]:
                    ii = np.arange(int(g["i0"]), int(g["i1"]))
                    jj = np.arange(int(g["j0"]), int(g["j1"]))
                    blk = scratch[int(g["buf_off"]) + ii[:, None] * int(g["ld"]) + jj[None, :]]
                    keep = np.ones(blk.shape, dtype=bool)
                    if g["lower"]:
                        keep = (int(g["diag_shift"]) + ii[:, None]) >= jj[None, :]
                    r = relpos[int(g["relrow_off"]) + ii] - int(t["drow_base"]) - int(t["row0"])
                    c = rlist[int(g["gcol_off"]) + jj] - int(t["dcol_base"]) - int(t["col0"])
                    assert r.min() >= 0 and r.max() < t["rows"] and c.min() >= 0 and c.max() < t["cols"]
                    sub = acc[np.ix_(r, c)]
                    sub[keep] += blk[keep]
                    acc[np.ix_(r, c)] = sub
                rows, cols = int(t["rows"]), int(t["cols"])
                idx = (int(t["d_off"]) + (int(t["row0"]) + np.arange(rows))[:, None] * int(t["d_ld"]) +
                       int(t["col0"]) + np.arange(cols)[None, :])
                arena[idx] -= acc[:rows, :cols]
            continue
        if kind == 0:
            for q in potrf[first:first + count]:
                n, ld, off = int(q["n"]), int(q["ld"]), int(q["off"])
                idx = off + np.arange(n)[:, None] * ld + np.arange(n)[None, :]
                blk = np.tril(arena[idx])
                Lb = blk if (q["flags"] & 1) else sl.cholesky(blk + np.tril(blk, -1).T, lower=True)
                if not (q["flags"] & 1):
                    low = np.tril_indices(n)
                    arena[idx[low]] = Lb[low]
                X = sl.solve_triangular(Lb, np.eye(n), lower=True)
                dinv[int(q["dinv_off"]):int(q["dinv_off"]) + n * n] = X.ravel()
            continue
        T = int(tile)
        for t in tiles[first:first + count]:
            u = units[int(t["unit"])]
            i0, j0 = int(t["ti"]) * T, int(t["tj"]) * T
            mi, nj = min(T, int(u["M"]) - i0), min(T, int(u["N"]) - j0)
            if kind == 9:          # k_trsm_rows: 64 rows x ALL columns of the chain block per workgroup,
                assert u["mode"] == MODE_TRSM and j0 == 0 and T == 64      # solved against its factored diagonal block
                cw, ld, off = int(u["N"]), int(u["d_ld"]), int(u["d_off"])
                cs, rr = int(u["d_col0"]), int(u["d_row0"]) + i0
                assert int(u["k0"]) == cs and int(u["klen"]) == cw and int(u["d_row0"]) == cs + cw
                dd = off + np.arange(cs, cs + cw)[:, None] * ld + np.arange(cs, cs + cw)[None, :]
                xi = off + np.arange(rr, rr + mi)[:, None] * ld + np.arange(cs, cs + cw)[None, :]
                arena[xi] = sl.solve_triangular(np.tril(arena[dd]), arena[xi].T, lower=True).T
                continue
            assert mi > 0 and nj > 0
            P = np.zeros((mi, nj))
            for sg in range(int(u["nseg"])):
                Ablk = seg_rows(u, sg, int(u["src_r0"]) + i0, mi, False)
                if u["mode"] == MODE_TRSM:
                    # B = the unit's N x dinv_ld matrix in the dinv scratch (Winv of a panel)
                    n, ldw = int(u["N"]), int(u["dinv_ld"])
                    X = dinv[int(u["dinv_off"]):int(u["dinv_off"]) + n * ldw].reshape(n, ldw)
                    assert Ablk.shape[1] == ldw
                    Bblk = X[j0:j0 + nj, :]
                else:
                    Bblk = seg_rows(u, sg, int(u["src_c0"]) + j0, nj, True)
                P += Ablk @ Bblk.T
            ii = np.arange(mi)[:, None] + i0
            jj = np.arange(nj)[None, :] + j0
            keep = np.ones((mi, nj), dtype=bool)
            if u["lower"]:
                keep = (int(u["src_r0"]) + ii) >= (int(u["src_c0"]) + jj)
            if u["mode"] == MODE_SCATTER:
                dr = relpos[int(u["relrow_off"]) + ii] - int(u["d_row0"])
                dc = rlist[int(u["gcol_off"]) + jj] - int(u["d_col0"])
                assert (dr[keep.any(axis=1)] >= 0).all() and (dc >= 0).all() and (dc < u["d_ld"]).all()
                idx = int(u["d_off"]) + dr * int(u["d_ld"]) + dc
                np.subtract.at(arena, idx[keep], P[keep])
            else:
                idx = (int(u["d_off"]) + (int(u["d_row0"]) + ii) * int(u["d_ld"]) +
                       int(u["d_col0"]) + jj)
                if u["mode"] == MODE_TRSM:
                    arena[idx] = P
                elif u["mode"] == 3:   # MODE_BUFFER: the product goes to the scratch block
                    sidx = int(u["d_off"]) + ii * int(u["d_ld"]) + jj
                    scratch[sidx] = P
                else:
                    arena[idx[keep]] -= P[keep]
    return arena


def emulate_solve(f, arena, y, job=0, phase=-1):
    """Numpy interpreter of the substitution program (spllt_hip_program_get
    "solve_*"): y is in pivot order, (nrhs, n), modified in place.  phase as
    spllt_hip_solve_dev: -1 everything, 0/1/2 the phases of a partitioned solve."""
    units, lst, tiles = f.program("solve_units"), f.program("solve_list"), f.program("solve_tiles")
    fwd, bwd = f.program("solve_fwd"), f.program("solve_bwd")
    nsub, ntop = (int(v) for v in f.program("solve_split"))
    rlist = f.sym("rlist")
    SR = 64  # kSolveStripRows

    def blk(u):
        w, nr, off = int(u["w"]), int(u["nrow"]), int(u["off"])
        return arena[off:off + nr * w].reshape(nr, w), rlist[int(u["idx_off"]):int(u["idx_off"]) + nr], w

    def run(launches):
        for kind, _lev, first, count in launches:
            if kind in (0, 3):       # DIAG forward / backward
                for b in lst[first:first + count]:
                    B, idx, w = blk(units[int(b)])
                    Ld = np.tril(B[:w])
                    y[:, idx[:w]] = sl.solve_triangular(Ld, y[:, idx[:w]].T, lower=True,
                                                        trans="N" if kind == 0 else "T").T
            else:                    # STRIP forward (1) / backward (2)
                for t in tiles[first:first + count]:
                    B, idx, w = blk(units[int(t["unit"])])
                    r0 = w + int(t["ti"]) * SR
                    r1 = min(r0 + SR, B.shape[0])
                    if kind == 1:
                        y[:, idx[r0:r1]] -= y[:, idx[:w]] @ B[r0:r1].T
                    else:
                        y[:, idx[:w]] -= y[:, idx[r0:r1]] @ B[r0:r1]
    do_f, do_b = job in (0, 1), job in (0, 2)
    if do_f and phase in (-1, 0):
        run(fwd[:nsub])
    if do_f and phase in (-1, 1):
        run(fwd[nsub:])
    if do_b and phase in (-1, 1):
        run(bwd[:ntop])
    if do_b and phase in (-1, 2):
        run(bwd[ntop:])
    return y


class LockstepExchange:
    """The collectives of a partitioned program for `world` emulated (or single-device) ranks
    that run in threads of this process: rank r calls ex.callback(r)(k, xbuf) at its exchange k,
    all ranks meet there, and everybody gets back what the collective of
    spllt_amd.multigpu.run_exchange would leave in its buffer.  Strict: the parts of a buffer a
    rank has no right to read afterwards (other ranks' chunks of a reduce-scatter, anything
    outside the broadcast segments) come back as NaN."""

    def __init__(self, world, plan):
        import threading
        self.world, self.plan = world, plan
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.out = [None] * world

    def _collective(self, k):
        kind, elems, chunk, segs = self.plan[k]
        bufs = self.slots
        for r in range(self.world):
            res = np.full_like(bufs[r], np.nan)
            if kind in (0, 3):
                res[:elems] = np.sum([b[:elems] for b in bufs], axis=0)
            elif kind == 1:      # the region of this level's reduce-scatter ends at elems
                lo = elems - self.world * chunk + r * chunk
                res[lo:lo + chunk] = np.sum([b[lo:lo + chunk] for b in bufs], axis=0)
            else:
                for root, off, cnt in segs:
                    res[off:off + cnt] = bufs[root][off:off + cnt]
            self.out[r] = res

    def callback(self, rank):
        def exchange(k, xbuf):
            self.slots[rank] = np.array(xbuf, copy=True)
            if self.bar.wait() == 0:
                self._collective(k)
            self.bar.wait()
            return self.out[rank]
        return exchange


def emulate_ranks(fs, val):
    """every rank's program of one partitioned factorization, interpreted in lockstep (one thread
    per rank); returns the ranks' arenas"""
    import threading
    from spllt_amd import multigpu
    ex = LockstepExchange(len(fs), multigpu.exchange_plan(fs[0]))
    res, err = [None] * len(fs), []

    def work(r):
        try:
            res[r] = emulate_program(fs[r], val, exchange=ex.callback(r), partitioned=True)
        except BaseException as e:   # noqa: BLE001 - surfaced below; do not leave the others waiting
            err.append(e)
            ex.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(fs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return res
