"""spllt_hip_analyse_symbolic: the analyse entry that takes the symbolic factorization
SpLLT gets from SSIDS (sptr, sparent, rptr, rlist, order; reference
src/spllt_analyse_mod.F90:129-158) instead of computing its own.  Feeding the product's own
exported quintuple back must reproduce the identical Symbolic and program."""
import numpy as np
import pytest

from emulate import emulate_program
from helpers import dense_arena, lower_mask, make_case, rel_err
from spllt_amd import api, matgen

CASES = [(lambda: matgen.poisson2d(20), 8, 4), (lambda: matgen.nd_like((7, 6, 6), 2), 32, 8),
         (lambda: matgen.fe27((4, 4, 3), 3), 48, 16), (lambda: matgen.poisson3d(8), 256, 32)]
I32 = ("order", "sptr", "sparent", "rlist", "small", "level", "bcol_node", "bcol_width", "bcol_r0",
       "bcol_nrow", "node_bcol0")
I64 = ("rptr", "bcol_off", "map_dst", "map_src", "lmap_ptr", "weight")


@pytest.mark.parametrize("gen,nb,nemin", CASES)
@pytest.mark.parametrize("prune", [False, True])
def test_own_quintuple_round_trips(gen, nb, nemin, prune):
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin, prune=prune, ncpu=3)
    quint = {k: f.sym(k) for k in ("sptr", "sparent", "rptr", "rlist", "order")}
    n, ptr, row, _ = api.csc_lower_1based(A)
    g = api.Factorization(n, ptr, row, nb=nb, nemin=1, prune_tree=prune, ncpu=3, symbolic=quint)
    assert g.sym_info()["ordering"] == "symbolic"
    for k in I32 + I64:
        assert np.array_equal(f.sym(k), g.sym(k)), k
    fi, gi = f.sym_info(), g.sym_info()
    for k in ("nnodes", "nbcol", "nblk", "arena", "nnz_l", "flops", "maxmn", "maxdepth"):
        assert fi[k] == gi[k], k
    for k in ("launches", "units", "tiles", "chains", "relpos"):
        assert np.array_equal(f.program(k), g.program(k)), k
    got = emulate_program(g, val)
    assert rel_err(got, dense_arena(g, A), lower_mask(g)) < 1e-13


def test_foreign_partition_is_taken_as_is():
    """a different (finer) supernode partition of the same elimination tree: every column its
    own node.  No amalgamation must happen, and the factor must still be right."""
    A = matgen.poisson2d(9)
    f, val = make_case(A, nb=8, nemin=1)
    n = f.n
    sptr, sparent, rptr, rlist, order = (f.sym(k) for k in ("sptr", "sparent", "rptr", "rlist", "order"))
    # split every supernode into single columns: column j's rows = rows of its node from j on
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
    quint = dict(sptr=np.array(nsptr), sparent=np.array(nspar), rptr=np.array(nrptr),
                 rlist=np.array(nrl), order=order)
    _, ptr, row, _ = api.csc_lower_1based(A)
    g = api.Factorization(n, ptr, row, nb=8, nemin=32, symbolic=quint)
    assert g.sym_info()["nnodes"] == n
    got = emulate_program(g, val)
    assert rel_err(got, dense_arena(g, A), lower_mask(g)) < 1e-13


@pytest.mark.parametrize("what", ["order", "postorder", "rows", "cover"])
def test_invalid_symbolic_input_is_a_parameter_error(what):
    A = matgen.poisson2d(8)
    f, _ = make_case(A, nb=8, nemin=4)
    q = {k: f.sym(k).copy() for k in ("sptr", "sparent", "rptr", "rlist", "order")}
    if what == "order":
        q["order"][0] = q["order"][1]                 # not a permutation
    elif what == "postorder":
        q["sparent"][0] = 0                           # parent must come after the child
    elif what == "rows":
        s = int(np.argmax(np.diff(q["rptr"]) - np.diff(q["sptr"]) >= 2))
        a = q["rptr"][s] + (q["sptr"][s + 1] - q["sptr"][s])
        q["rlist"][a], q["rlist"][a + 1] = q["rlist"][a + 1], q["rlist"][a]   # unsorted
    else:
        s = int(np.argmax(np.diff(q["rptr"]) - np.diff(q["sptr"]) >= 1))
        # drop the last row of a node: the pattern of A (or of a child) is no longer covered
        keep = np.ones(len(q["rlist"]), dtype=bool)
        keep[q["rptr"][s + 1] - 1] = False
        q["rlist"] = q["rlist"][keep]
        q["rptr"][s + 1:] -= 1
    n, ptr, row, _ = api.csc_lower_1based(A)
    with pytest.raises(api.SplltError) as e:
        api.Factorization(n, ptr, row, nb=8, symbolic=q)
    assert e.value.flag == -10
