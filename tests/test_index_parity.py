"""Index parity of the scatter maps: the reference recomputes
row_list / col_list per update_between task with spllt_update_between_compute_map
(src/spllt_kernels_mod.F90:1606-1723); the product precomputes them once at analyse time
(Program::relpos, the source row list as gcol, the unit's row / column ranges).  Integer
work: the product's lists must equal the oracle's restatement (oracle/spllt_oracle.c,
spo_compute_map) entry for entry, for every SCATTER unit and every destination tile."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_case
from oracle import pyoracle
from spllt_amd import matgen

CASES = [
    ("p2d20-nb8", lambda: matgen.poisson2d(20), 8, 4),
    ("p3d8-nb16", lambda: matgen.poisson3d(8), 16, 4),
    ("box7-nb24", lambda: matgen.nd_like((7, 7, 6), 2), 24, 8),
    ("fe27-nb40", lambda: matgen.fe27((5, 4, 4), 3), 40, 16),
    ("box9-nb256", lambda: matgen.nd_like((9, 9, 8), 2), 256, 32),
]


@pytest.mark.parametrize("name,gen,nb,nemin", CASES)
@pytest.mark.parametrize("flags", [0, 64])
def test_scatter_maps_equal_compute_map(name, gen, nb, nemin, flags):
    f, _ = make_case(gen(), nb=nb, nemin=nemin, engine_flags=flags)
    olib = pyoracle.load("plain")
    units, relpos = f.program("units"), f.program("relpos")
    sptr, rptr, rlist = f.sym("sptr"), f.sym("rptr"), f.sym("rlist")
    bc_off, bc_node, bc_nrow = f.sym("bcol_off"), f.sym("bcol_node"), f.sym("bcol_nrow")
    node_bc0 = f.sym("node_bcol0")
    off2bcol = {int(o): b for b, o in enumerate(bc_off)}
    ip = C.POINTER(C.c_int)
    nscat = 0
    for u in units:
        if u["mode"] != 1:
            continue
        nscat += 1
        s = int(bc_node[int(u["src_bcol0"])])
        db = off2bcol[int(u["d_off"])]
        a = int(bc_node[db])
        dcol = db - int(node_bc0[a])
        s_index = np.ascontiguousarray(rlist[rptr[s]:rptr[s + 1]], dtype=np.int32)
        d_index = np.ascontiguousarray(rlist[rptr[a]:rptr[a + 1]], dtype=np.int32)
        M, N, r0, c0 = int(u["M"]), int(u["N"]), int(u["src_r0"]), int(u["src_c0"])
        assert r0 == c0 and r0 + M == len(s_index)
        my_cols = rlist[int(u["gcol_off"]):int(u["gcol_off"]) + N] - int(u["d_col0"])
        my_rows = relpos[int(u["relrow_off"]):int(u["relrow_off"]) + M] - int(u["d_row0"])
        assert int(u["d_row0"]) == dcol * nb
        ntile = (int(bc_nrow[db]) + nb - 1) // nb
        covered = []
        for t in range(ntile):
            row_list = np.zeros(nb, dtype=np.int32)
            col_list = np.zeros(nb, dtype=np.int32)
            out = [C.c_int(0) for _ in range(6)]
            # scol: any source block column gives the same lists (the merge starts at or
            # before the node's first off-diagonal row); use the unit's own first one
            scol = int(u["src_bcol0"]) - int(node_bc0[s])
            ok = olib.spo_compute_map(int(sptr[a]), int(sptr[a + 1]) - 1, nb, d_index.ctypes.data_as(ip),
                                      len(d_index), dcol, t, int(sptr[s]), int(sptr[s + 1]) - 1, nb,
                                      s_index.ctypes.data_as(ip), len(s_index), scol,
                                      row_list.ctypes.data_as(ip), col_list.ctypes.data_as(ip),
                                      *[C.byref(v) for v in out])
            rls, cls, s1sa, s1en, s2sa, s2en = (v.value for v in out)
            if not ok or rls == 0:
                continue
            # column side: the unit's column range and list are the reference's
            assert (s1sa, s1en) == (c0, c0 + N - 1)
            assert cls == N and np.array_equal(col_list[:cls], my_cols)
            # row side: rows s2sa..s2en of the source land in tile t at row_list
            assert s2en - s2sa + 1 == rls
            assert np.array_equal(row_list[:rls] + t * nb, my_rows[s2sa - r0:s2en - r0 + 1])
            covered.extend(range(s2sa, s2en + 1))
        # every row of the unit is scattered into exactly one destination tile
        assert covered == list(range(r0, r0 + M))
    assert nscat > 0
    f.close()
