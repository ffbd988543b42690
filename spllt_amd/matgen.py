"""Deterministic SPD test matrices for the BASELINE.json configurations
 and geometric nested-dissection orderings for
them.  No SuiteSparse file can be fetched in this environment, so configs 2, 4
and 5 use structural stand-ins; a real MatrixMarket file is used when a path is
given (read_mtx).  All values are diagonally dominant, like the matrices the
reference's drivers factorize (rb_options%values = 3, drivers/spllt_omp.F90:81).
"""
import numpy as np
import scipy.sparse as sp


def _grid_ids(shape):
    return np.arange(int(np.prod(shape)), dtype=np.int64).reshape(shape)


def stencil_matrix(shape, radius=1, star=True, dof=1):
    """Symmetric matrix of a box/star stencil on a regular grid.

    star=True : neighbours along the axes only (5-pt / 7-pt for radius 1);
    star=False: full box stencil ((2r+1)^d - 1 neighbours).
    Off-diagonals are -1; the diagonal is the classical Poisson value
    (2*d) for the radius-1 star stencil, else 1 + sum|offdiag| (strictly
    diagonally dominant).  dof>1 couples `dof` unknowns per vertex densely.
    Returns scipy CSC (full symmetric)."""
    shape = tuple(int(s) for s in shape)
    d = len(shape)
    ids = _grid_ids(shape)
    rows, cols = [], []
    offs = []
    rng = range(-radius, radius + 1)
    import itertools
    for off in itertools.product(rng, repeat=d):
        if all(o == 0 for o in off):
            continue
        if star and sum(1 for o in off if o != 0) != 1:
            continue
        if off > tuple([0] * d):  # one direction of each pair
            offs.append(off)
    for off in offs:
        sl_a, sl_b = [], []
        for ax, o in enumerate(off):
            if o >= 0:
                sl_a.append(slice(0, shape[ax] - o))
                sl_b.append(slice(o, shape[ax]))
            else:
                sl_a.append(slice(-o, shape[ax]))
                sl_b.append(slice(0, shape[ax] + o))
        a = ids[tuple(sl_a)].ravel()
        b = ids[tuple(sl_b)].ravel()
        rows.append(a)
        cols.append(b)
    nv = ids.size
    r = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    c = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    if dof > 1:
        # vertex graph -> dof x dof dense blocks (including the vertex's own block)
        vr = np.concatenate([r, np.arange(nv)])
        vc = np.concatenate([c, np.arange(nv)])
        di, dj = np.meshgrid(np.arange(dof), np.arange(dof), indexing="ij")
        R = (vr[:, None] * dof + di.ravel()[None, :]).ravel()
        Cc = (vc[:, None] * dof + dj.ravel()[None, :]).ravel()
        keep = R != Cc
        r, c = R[keep], Cc[keep]
        # keep one direction of every pair
        lo = np.minimum(r, c)
        hi = np.maximum(r, c)
        key = lo * (nv * dof) + hi
        _, idx = np.unique(key, return_index=True)
        r, c = lo[idx], hi[idx]
        nv = nv * dof
    off = sp.coo_matrix((-np.ones(r.size), (r, c)), shape=(nv, nv))
    A = (off + off.T).tocsc()
    if star and radius == 1 and dof == 1:
        diag = np.full(nv, 2.0 * d)
    else:
        diag = 1.0 + np.asarray(abs(A).sum(axis=1)).ravel()
    A = (A + sp.diags(diag)).tocsc()
    A.sort_indices()
    return A


def poisson2d(n):
    """5-point Poisson on an n x n grid (config 1: n = 128)."""
    return stencil_matrix((n, n), 1, True)


def poisson3d(n):
    """7-point Poisson on an n^3 grid (config 3: n = 128)."""
    return stencil_matrix((n, n, n), 1, True)


def nd_like(shape=(42, 42, 41), radius=3):
    """Stand-in for SuiteSparse ND/nd24k (config 2): radius-3 box stencil,
    342 neighbours per interior row, 72 324 dofs for the default shape."""
    return stencil_matrix(shape, radius, False)


def fe27(shape, dof=3):
    """Stand-in for Flan_1565 / Serena / audikw_1 (configs 4, 5): 27-point
    stencil with `dof` unknowns per vertex."""
    return stencil_matrix(shape, 1, False, dof)


def geometric_nd_order(shape, radius=1, dof=1, leaf=64):
    """Nested dissection by recursive coordinate bisection for a grid stencil
    of the given radius: the separator of a box is the middle slab (thickness
    `radius`) orthogonal to its longest edge.  Returns order[var] = 1-based
    pivot position (the convention of the C-ABI `order` array)."""
    shape = tuple(int(s) for s in shape)
    ids = _grid_ids(shape)
    seq = []

    def rec(lo, hi):
        ext = [h - l for l, h in zip(lo, hi)]
        if min(ext) <= 0:
            return
        nv = int(np.prod(ext))
        ax = int(np.argmax(ext))
        if nv <= leaf or ext[ax] <= 2 * radius:
            seq.append(ids[tuple(slice(l, h) for l, h in zip(lo, hi))].ravel())
            return
        mid = lo[ax] + (ext[ax] - radius) // 2
        a_hi = list(hi)
        a_hi[ax] = mid
        b_lo = list(lo)
        b_lo[ax] = mid + radius
        rec(lo, a_hi)
        rec(b_lo, hi)
        s_lo, s_hi = list(lo), list(hi)
        s_lo[ax], s_hi[ax] = mid, mid + radius
        seq.append(ids[tuple(slice(l, h) for l, h in zip(s_lo, s_hi))].ravel())

    rec([0] * len(shape), list(shape))
    perm = np.concatenate(seq)  # perm[pos] = vertex
    if dof > 1:
        perm = (perm[:, None] * dof + np.arange(dof)[None, :]).ravel()
    order = np.empty(perm.size, dtype=np.int32)
    order[perm] = np.arange(1, perm.size + 1, dtype=np.int32)
    return order


def read_mtx(path):
    """MatrixMarket coordinate reader (real/integer/pattern, symmetric or
    general).  Pattern-only files get invented values like the reference does
    (src/spllt_mod.F90:480-485); every matrix is then made diagonally dominant
    like rb_options%values = 3 when `dominant` would be needed by the caller."""
    import scipy.io
    A = sp.csc_matrix(scipy.io.mmread(path))
    if A.shape[0] != A.shape[1]:
        raise ValueError("matrix is not square")
    A = ((A + A.T) * 0.5).tocsc()
    return A


def invent_values(count, seed=1):
    """Values for a pattern-only file (the reference makes them up with SPRAL's random_real,
    src/spllt_mod.F90:480-485): uniform in (-1, 1) from splitmix64 -- the stream the library's
    readers use (spllt_amd/csrc/readers.cpp), so that both give the same matrix."""
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, count + 1, dtype=np.uint64)
        z = (s ^ (s >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (2.0 / 9007199254740992.0) - 1.0


def read_file_c(path, kind=None, values=3, seed=1):
    """A matrix file through the LIBRARY's readers (spllt_hip_read_rb / spllt_hip_read_mm, the
    C boundary): returns (n, ptr, row, val) of the lower triangle, 1-based CSC, as numpy arrays."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    if kind is None:
        kind = "mm" if str(path).endswith((".mtx", ".mm")) else "rb"
    n, nnz = C.c_int(), C.c_int()
    ptr, row, val = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    fn = lib.spllt_hip_read_mm if kind == "mm" else lib.spllt_hip_read_rb
    rc = fn(str(path).encode(), values, seed, C.byref(n), C.byref(nnz), C.byref(ptr), C.byref(row), C.byref(val))
    if rc != 0:
        raise ValueError(f"{path}: reader failed with flag {rc}")
    try:
        p = np.ctypeslib.as_array(ptr, shape=(n.value + 1,)).copy()
        r = np.ctypeslib.as_array(row, shape=(max(nnz.value, 1),))[:nnz.value].copy()
        v = np.ctypeslib.as_array(val, shape=(max(nnz.value, 1),))[:nnz.value].copy()
    finally:
        lib.spllt_hip_free_matrix(ptr, row, val)
    return n.value, p, r, v


def _fortran_fields(fmt):
    """(count, width) of a Rutherford-Boeing Fortran format such as (10I8),
    (1P,3E25.16), (4D20.12) or (8F10.3): fields are fixed-width and may abut"""
    import re
    m = re.search(r"(\d+)\s*[IiEeDdFfGg]\s*(\d+)", fmt.replace(" ", ""))
    if not m:
        raise ValueError(f"unsupported Fortran format {fmt!r}")
    return int(m.group(1)), int(m.group(2))


def _read_fixed(lines, pos, nlines, fmt, count, conv):
    per, width = _fortran_fields(fmt)
    out = []
    for ln in lines[pos:pos + nlines]:
        ln = ln.rstrip("\n")
        for k in range(per):
            if len(out) == count:
                break
            fld = ln[k * width:(k + 1) * width]
            if fld.strip():
                out.append(conv(fld))
    if len(out) != count:
        raise ValueError(f"expected {count} entries, found {len(out)}")
    return out


def read_rb(path, values=3, seed=1):
    """Rutherford-Boeing reader for assembled symmetric matrices (type ?sa: rsa / psa /
    isa), the 'csc' input of the reference's drivers (drivers/spllt_omp.F90:78-85, SPRAL
    rb_read).  `values` follows the driver's use of rb_options%values:
      0  values as in the file (a pattern-only file is an error),
      3  what every reference driver asks for ("force diagonal dominance"): keep the
         file's off-diagonal values, or invent them (uniform in (-1, 1), invent_values --
         the reference uses SPRAL's random_real for missing values,
         src/spllt_mod.F90:480-485) when the file holds a pattern only, then set every
         diagonal entry to 1 + sum |off-diagonal entries of its row| so that the matrix
         is strictly diagonally dominant, hence positive definite.
    Returns the full symmetric matrix as scipy CSC."""
    with open(path) as fh:
        lines = fh.readlines()
    if len(lines) < 4:
        raise ValueError("not a Rutherford-Boeing file")
    cards = lines[1].split()
    totcrd, ptrcrd, indcrd = int(cards[0]), int(cards[1]), int(cards[2])
    valcrd = int(cards[3]) if len(cards) > 3 else totcrd - ptrcrd - indcrd
    l3 = lines[2].split()
    mxtype = l3[0].lower()
    nrow, ncol, nnz = int(l3[1]), int(l3[2]), int(l3[3])
    if len(mxtype) != 3 or mxtype[1] != "s" or mxtype[2] != "a" or nrow != ncol:
        raise ValueError(f"only assembled symmetric matrices (?sa) are supported, got {mxtype!r}")
    fm = lines[3]
    ptrfmt, indfmt = fm[:16], fm[16:32]
    valfmt = fm[32:52] if len(fm) > 32 else ""
    pos = 4
    ptr = np.array(_read_fixed(lines, pos, ptrcrd, ptrfmt, ncol + 1, int), dtype=np.int64) - 1
    pos += ptrcrd
    ind = np.array(_read_fixed(lines, pos, indcrd, indfmt, nnz, int), dtype=np.int64) - 1
    pos += indcrd
    has_values = mxtype[0] in "ri" and valcrd > 0
    if has_values:
        val = np.array(_read_fixed(lines, pos, valcrd, valfmt, nnz,
                                   lambda t: float(t.replace("D", "E").replace("d", "e"))))
    elif values == 0:
        raise ValueError("pattern-only file and values=0")
    else:
        val = invent_values(nnz, seed)
    if ptr[0] != 0 or ptr[-1] != nnz or (np.diff(ptr) < 0).any() or ind.min() < 0 or ind.max() >= nrow:
        raise ValueError("inconsistent column pointers / row indices")
    L = sp.csc_matrix((val, ind, ptr), shape=(nrow, ncol))
    L = sp.tril(L, format="csc")                  # stored triangle (lower by convention)
    A = (L + sp.tril(L, -1).T).tocsc()
    if values == 3:
        off = (A - sp.diags(A.diagonal())).tocsc()
        diag = 1.0 + np.asarray(abs(off).sum(axis=1)).ravel()
        A = (off + sp.diags(diag)).tocsc()
    A.sort_indices()
    return A


def write_rb(path, A, pattern_only=False, title="spllt-hip test matrix", key="SPLLTHIP"):
    """Writes the lower triangle of a symmetric matrix as an rsa / psa Rutherford-Boeing
    file (used by the tests of read_rb)."""
    L = sp.tril(sp.csc_matrix(A), format="csc")
    L.sort_indices()
    n, nnz = L.shape[0], L.nnz

    def cards(vals, per, fmt):
        vals = list(vals)
        return ["".join(fmt % v for v in vals[i:i + per]) for i in range(0, len(vals), per)]
    pl = cards(L.indptr + 1, 8, "%10d")
    il = cards(L.indices + 1, 8, "%10d")
    vl = [] if pattern_only else cards(L.data, 3, "%26.17E")
    with open(path, "w") as fh:
        fh.write(f"{title:<72}{key:<8}\n")
        fh.write(f"{len(pl) + len(il) + len(vl):14d}{len(pl):14d}{len(il):14d}{len(vl):14d}\n")
        fh.write(f"{'psa' if pattern_only else 'rsa':<14}{n:14d}{n:14d}{nnz:14d}{0:14d}\n")
        fh.write(f"{'(8I10)':<16}{'(8I10)':<16}{'' if pattern_only else '(3E26.17)':<20}\n")
        for ln in pl + il + vl:
            fh.write(ln + "\n")


def make_diag_dominant(A):
    """Replace the values by a diagonally dominant set on the same pattern
    (what SPRAL's rb_read does for values=3, as used by every reference driver)."""
    A = sp.csc_matrix(A)
    off = A - sp.diags(A.diagonal())
    off.data = -np.ones_like(off.data)
    diag = 1.0 + np.asarray(abs(off).sum(axis=1)).ravel()
    B = (off + sp.diags(diag)).tocsc()
    B.sort_indices()
    return B


CONFIGS = {
    # name: (generator, kwargs, nb, ordering args) -- SURVEY.md 8(d)
    "poisson2d_128": dict(gen="poisson2d", shape=(128, 128), radius=1, dof=1, nb=256),
    "nd24k_like": dict(gen="nd_like", shape=(42, 42, 41), radius=3, dof=1, nb=256),
    "poisson3d_128": dict(gen="poisson3d", shape=(128, 128, 128), radius=1, dof=1, nb=384),
    "flan_like": dict(gen="fe27", shape=(80, 80, 81), radius=1, dof=3, nb=512),
    "serena_like": dict(gen="fe27", shape=(68, 68, 68), radius=1, dof=3, nb=768),
}


def build_config(name, scale=1.0):
    """Matrix + geometric order of a named configuration; `scale` shrinks every
    grid edge (used for bounded CPU-baseline samples and tests)."""
    cfg = dict(CONFIGS[name])
    shape = tuple(max(2 * cfg["radius"] + 2, int(round(s * scale))) for s in cfg["shape"])
    if cfg["gen"] in ("poisson2d", "poisson3d"):
        A = stencil_matrix(shape, 1, True)
    elif cfg["gen"] == "nd_like":
        A = stencil_matrix(shape, cfg["radius"], False)
    else:
        A = stencil_matrix(shape, 1, False, cfg["dof"])
    order = geometric_nd_order(shape, cfg["radius"], cfg["dof"])
    cfg["shape"] = shape
    return A, order, cfg
