"""Shared helpers of the test-suite (CPU side)."""
import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp

from oracle import pyoracle
from spllt_amd import api, matgen  # noqa: F401


def sym_tables(f):
    return {k: f.sym(k) for k in ("order", "sptr", "sparent", "rptr", "rlist", "bcol_off",
                                  "bcol_width", "bcol_r0", "bcol_nrow", "bcol_node", "node_bcol0",
                                  "level", "small")}


def lower_mask(f):
    """Boolean mask over the arena: True on entries that hold L (row >= col);
    False on the never-read strict upper triangle of diagonal tiles
    (parity checks must mask it)."""
    t = sym_tables(f)
    mask = np.zeros(f.sym_info()["arena"], dtype=bool)
    for b in range(len(t["bcol_off"])):
        w, nr, off = int(t["bcol_width"][b]), int(t["bcol_nrow"][b]), int(t["bcol_off"][b])
        m = np.ones((nr, w), dtype=bool)
        m[:w, :w] = np.tril(np.ones((w, w), dtype=bool))
        mask[off:off + nr * w] = m.ravel()
    return mask


def dense_arena(f, A):
    """Expected arena from an independent dense LAPACK Cholesky of P A P^T."""
    t = sym_tables(f)
    n = f.n
    P = np.empty(n, dtype=np.int64)
    P[t["order"]] = np.arange(n)
    Ld = sl.cholesky(sp.csc_matrix(A).toarray()[np.ix_(P, P)], lower=True)
    arena = np.zeros(f.sym_info()["arena"])
    for b in range(len(t["bcol_off"])):
        s = int(t["bcol_node"][b])
        rows = t["rlist"][t["rptr"][s]:t["rptr"][s + 1]]
        w, nr, off, r0 = (int(t["bcol_width"][b]), int(t["bcol_nrow"][b]), int(t["bcol_off"][b]),
                          int(t["bcol_r0"][b]))
        c0 = int(t["sptr"][s]) + r0
        blk = Ld[np.ix_(rows[r0:r0 + nr], np.arange(c0, c0 + w))]
        arena[off:off + nr * w] = blk.ravel()
    return arena


def make_case(A, nb, nemin=32, prune=False, ncpu=1, order=None, **kw):
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=nb, nemin=nemin, prune_tree=prune, ncpu=ncpu,
                          order=order, **kw)
    return f, val


def oracle_factor(f, val, variant="plain", nthreads=1, use_small=False, min_width_blas=8):
    small = f.sym("small") if use_small else None
    o = pyoracle.OracleFactor.from_factorization(f, small=small, variant=variant,
                                                 min_width_blas=min_width_blas)
    rc = o.factorize(val, nthreads)
    return o, rc


def rel_err(a, b, mask=None):
    if mask is not None:
        a, b = a[mask], b[mask]
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def bwd_err(A, x, b):
    """Scaled backward error of the reference's checker
    (check_backward_error_multi, src/utils_mod.F90:432-478):
    ||b - A x||_2 / (||b||_2 + max|a_ij| ||x||_2); pass iff <= 1e-14."""
    r = b - A @ x
    return float(np.linalg.norm(r) / (np.linalg.norm(b) + abs(A).max() * np.linalg.norm(x)))


def drive_exchanges(fs, bufs, on_exchange=None):
    """All exchanges of a partitioned factorization whose rank-engines run in THIS process on
    one device (fs[r] = rank r, bufs[r] = its torch exchange buffer): every engine has been
    started (factor_dev); at each exchange point the streams are drained, the collective is
    done with torch ops (what spllt_amd.multigpu.run_exchange does across devices), and the
    engines continue.  Strict: what a rank may not read after a collective comes back as NaN.
    Returns the number of exchanges."""
    import torch
    from spllt_amd import multigpu
    plan = multigpu.exchange_plan(fs[0])
    world, nx = len(fs), 0
    while True:
        ks = [f.pending_exchange() for f in fs]
        assert len(set(ks)) == 1, ks          # every rank is at the same exchange
        k = ks[0]
        if k < 0:
            return nx
        for f in fs:
            f.wait()                           # at an exchange point: drains the streams only
        kind, elems, chunk, segs = plan[k]
        src = [b.clone() for b in bufs]
        if on_exchange:
            on_exchange(k, kind, src)
        for r, b in enumerate(bufs):
            b.fill_(float("nan"))
            if kind in (0, 3):
                b[:elems] = torch.stack([s[:elems] for s in src]).sum(dim=0)
            elif kind == 1:      # the region of this level's reduce-scatter ends at elems
                lo = elems - world * chunk + r * chunk
                b[lo:lo + chunk] = torch.stack([s[lo:lo + chunk] for s in src]).sum(dim=0)
            else:
                for root, off, cnt in segs:
                    b[off:off + cnt] = src[root][off:off + cnt]
        torch.cuda.synchronize()
        for f in fs:
            f.continue_after_exchange()
        nx += 1


# ---- foreign symbolic factorizations for spllt_hip_analyse_symbolic ----------
def quintuple_single_columns(f):
    """every column its own supernode: the finest partition of f's elimination tree
    (sptr, sparent, rptr, rlist, order as SSIDS would deliver them, 0-based)"""
    n = f.n
    sptr, sparent, rptr, rlist, order = (f.sym(k) for k in ("sptr", "sparent", "rptr", "rlist", "order"))
    nsptr, nspar, nrptr, nrl = [0], [], [0], []
    for s in range(len(sparent)):
        rows = rlist[rptr[s]:rptr[s + 1]]
        nc = sptr[s + 1] - sptr[s]
        for k in range(nc):
            j = sptr[s] + k
            nsptr.append(j + 1)
            nrl.extend(rows[k:].tolist())
            nrptr.append(len(nrl))
            if k + 1 < nc:
                nspar.append(j + 1)
            else:
                p = sparent[s]
                nspar.append(int(sptr[p]) if p < len(sparent) else n)
    return dict(sptr=np.array(nsptr), sparent=np.array(nspar), rptr=np.array(nrptr), rlist=np.array(nrl),
                order=order)


def quintuple_hand_amalgamated(f, every=2):
    """a coarser partition than f's: a node is merged into its parent where the parent follows it
    directly in the postorder and has no other child (columns stay contiguous; the merged row list
    = the two column sets, then the sorted union of the rows below -- explicit zeros where the
    child had fewer rows).  Every `every`-th eligible pair is merged, bottom-up."""
    sptr, sparent, rptr, rlist, order = (np.asarray(f.sym(k)) for k in ("sptr", "sparent", "rptr", "rlist", "order"))
    nn = len(sparent)
    nchild = np.zeros(nn + 1, dtype=int)
    for s in range(nn):
        nchild[min(sparent[s], nn)] += 1
    nodes = [dict(c0=int(sptr[s]), c1=int(sptr[s + 1]), rows=list(map(int, rlist[rptr[s]:rptr[s + 1]])),
                  parent=int(sparent[s]), alive=True) for s in range(nn)]
    k = 0
    for s in range(nn - 1):
        p = nodes[s]["parent"]
        if p != s + 1 or p >= nn or nchild[p] != 1:
            continue
        k += 1
        if k % every:
            continue
        a, b = nodes[s], nodes[p]
        below = sorted((set(a["rows"][a["c1"] - a["c0"]:]) | set(b["rows"][b["c1"] - b["c0"]:])) - set(range(b["c0"], b["c1"])))
        b["rows"] = list(range(a["c0"], b["c1"])) + below
        b["c0"] = a["c0"]
        a["alive"] = False
        nchild[p] = nchild[s]             # the merged node inherits the child's children
    newid = {}
    for s in range(nn):
        if nodes[s]["alive"]:
            newid[s] = len(newid)
    def target(s):                        # a dead node lives on in its parent
        while s < nn and not nodes[s]["alive"]:
            s = nodes[s]["parent"]
        return s
    nsptr, nspar, nrptr, nrl = [0], [], [0], []
    for s in range(nn):
        if not nodes[s]["alive"]:
            continue
        nsptr.append(nodes[s]["c1"])
        nrl.extend(nodes[s]["rows"])
        nrptr.append(len(nrl))
        p = target(nodes[s]["parent"]) if nodes[s]["parent"] < nn else nn
        nspar.append(newid[p] if p < nn else len(newid))
    return dict(sptr=np.array(nsptr), sparent=np.array(nspar), rptr=np.array(nrptr), rlist=np.array(nrl),
                order=order)
