"""Numpy interpreter of the stream-DAG program exported by
spllt_hip_program_get.  TEST-ONLY: it executes the same work tables the HIP
kernels consume (potrf units, update units, tiles, launches) with dense numpy
arithmetic, so that the host-side scheduler can be validated on a machine
without a GPU.  It is not part of the product and is never timed."""
import numpy as np
import scipy.linalg as sl

MODE_DIRECT, MODE_SCATTER, MODE_TRSM = 0, 1, 2


def emulate_program(f, val, exchange=None, partitioned=False):
    """exchange(xbuf) -> summed xbuf is called at the EXCHANGE marker of a
    partitioned (multi-GPU) program with the packed top-tree block columns."""
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

    for kind, level, first, count, tile, _fl, _st, _w0, _w1, _rec in launches:
        if kind == 2:  # EXCHANGE: pack the top-tree block columns, reduce, unpack
            top = f.partition("top_bcols")
            sl_ = [slice(int(bc_off[b]), int(bc_off[b]) + int(f.sym("bcol_nrow")[b]) * int(bc_w[b]))
                   for b in top]
            xbuf = np.concatenate([arena[s_] for s_ in sl_]) if sl_ else np.zeros(0)
            xbuf = exchange(xbuf)
            o = 0
            for s_ in sl_:
                k = s_.stop - s_.start
                arena[s_] = xbuf[o:o + k]
                o += k
            continue
        if kind == 4:  # single-workgroup panel chain = Cholesky of the diagonal tile + panel inverses
            for q in f.program("chains")[first:first + count]:
                w, nt, off, pwq = int(q["ld"]), int(q["n"]), int(q["off"]), int(q["flags"])
                idx = off + np.arange(nt)[:, None] * w + np.arange(nt)[None, :]
                blk = np.tril(arena[idx])
                Lb = sl.cholesky(blk + np.tril(blk, -1).T, lower=True)
                low = np.tril_indices(nt)
                arena[idx[low]] = Lb[low]
                slot = int(q["dinv_off"])
                for c0 in range(0, nt, pwq):
                    pn = min(pwq, nt - c0)
                    X = sl.solve_triangular(Lb[c0:c0 + pn, c0:c0 + pn], np.eye(pn), lower=True)
                    dinv[slot:slot + pn * pn] = X.ravel()
                    slot += pn * pn
            continue
        if kind == 3:  # fused strip TRSM: rows below the diagonal tile, all panels
            strips = f.program("strips")
            rs = int(tile)
            for t in tiles[first:first + count]:
                q = strips[int(t["unit"])]
                w, off = int(q["ld"]), int(q["off"])
                r0 = int(q["row0"]) + int(t["ti"]) * rs
                nr = min(rs, int(q["row0"]) + int(q["nrows"]) - r0)
                Lt = np.tril(arena[off:off + w * w].reshape(w, w))
                rows = arena[off + r0 * w: off + (r0 + nr) * w].reshape(nr, w)
                arena[off + r0 * w: off + (r0 + nr) * w] = sl.solve_triangular(
                    Lt, rows.T, lower=True).T.ravel()
            continue
        if kind == 5:  # fused panel step: TRSM of the rows below + update of the next panel
            panels = f.program("panels")
            tl = tiles[first:first + count]
            for t in tl:  # pass 1: X = A * inv(L_pp)^T
                q = panels[int(t["unit"])]
                ld, off, c0, pn = int(q["ld"]), int(q["off"]), int(q["c0"]), int(q["pn"])
                D = dinv[int(q["dinv_off"]):int(q["dinv_off"]) + pn * pn].reshape(pn, pn)
                r0 = c0 + pn + int(t["ti"]) * 32
                nr = min(32, c0 + pn + int(q["nrows"]) - r0)
                assert nr > 0
                idx = off + (r0 + np.arange(nr))[:, None] * ld + c0 + np.arange(pn)[None, :]
                arena[idx] = arena[idx] @ D.T
            for t in tl:  # pass 2: next panel -= [S | O | X]_i [S | O | X]_d^T
                q = panels[int(t["unit"])]
                if q["d_off"] < 0:
                    continue
                ld, off, c0, pn = int(q["ld"]), int(q["off"]), int(q["c0"]), int(q["pn"])
                rb, dpn = c0 + pn, int(q["d_pn"])
                i0 = int(t["ti"]) * 32
                nr = min(32, int(q["nrows"]) - i0)

                def rows_of(r_first, cnt):
                    parts = []
                    if q["s_off"] >= 0:
                        sld, sk, rsh = int(q["s_ld"]), int(q["s_k"]), int(q["s_rshift"])
                        ii = int(q["s_off"]) + (r_first + rsh + np.arange(cnt))[:, None] * sld + np.arange(sk)[None, :]
                        parts.append(arena[ii])
                    ii = off + (r_first + np.arange(cnt))[:, None] * ld + np.arange(c0 + pn)[None, :]
                    parts.append(arena[ii])
                    return np.hstack(parts)
                P = rows_of(rb + i0, nr) @ rows_of(rb, dpn).T
                ii = np.arange(nr)[:, None] + i0
                jj = np.arange(dpn)[None, :]
                keep = ii >= jj
                idx = (int(q["d_off"]) + (rb + ii - int(q["d_rshift"])) * int(q["d_ld"]) +
                       int(q["d_c0"]) + jj)
                arena[idx[keep]] -= P[keep]
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
            assert mi > 0 and nj > 0
            P = np.zeros((mi, nj))
            for sg in range(int(u["nseg"])):
                Ablk = seg_rows(u, sg, int(u["src_r0"]) + i0, mi, False)
                if u["mode"] == MODE_TRSM:
                    n = int(u["dinv_ld"])
                    X = dinv[int(u["dinv_off"]):int(u["dinv_off"]) + n * n].reshape(n, n)
                    Bblk = X[j0:j0 + nj, :Ablk.shape[1]]
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
