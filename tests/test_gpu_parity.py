"""GPU parity tests (run on the MI355X box): the HIP path, called through the
C-ABI of libspllt_hip.so, against the CPU oracle on identical symbolic input.

Tolerance (fp64): max |L_hip - L_oracle| / max|L_oracle| <= 1e-12 over the
lower-triangular (meaningful) entries -- summation order differs (MFMA K-order,
left-looking panels, TRSM through inverted 64x64 diagonal panels, fp64 atomics
in the scatter epilogue) so bit equality is not expected; and the reference's
own acceptance bar, scaled backward error <= 1e-14 (src/utils_mod.F90:462-467).
"""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import drive_exchanges, bwd_err, dense_arena, lower_mask, make_case, oracle_factor, rel_err
from spllt_amd import api, matgen

pytestmark = pytest.mark.gpu
TOL_L = 1e-12


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


CASES = [
    ("kat3", lambda: sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(3, 3)).tocsc(), 4, 32, None),
    ("p2d12-nb4", lambda: matgen.poisson2d(12), 4, 4, None),
    ("p2d16-nb8", lambda: matgen.poisson2d(16), 8, 4, None),
    ("p2d32-nb16", lambda: matgen.poisson2d(32), 16, 32, None),
    ("p2d64-nb100-pw32", lambda: matgen.poisson2d(64), 100, 32, 32),
    ("p3d10-nb48", lambda: matgen.poisson3d(10), 48, 16, None),
    ("box8-nb96", lambda: matgen.nd_like((8, 8, 8), 2), 96, 16, None),
    ("box12-nb256", lambda: matgen.nd_like((12, 12, 11), 3), 256, 32, None),
    ("fe27-nb64", lambda: matgen.fe27((5, 5, 4), 3), 64, 16, None),
    ("p2d128-nb256", lambda: matgen.poisson2d(128), 256, 32, None),  # BASELINE config 1
    ("p3d20-nb384", lambda: matgen.poisson3d(20), 384, 32, None),    # config 3 tile size
    ("fe27-nb512", lambda: matgen.fe27((9, 8, 8), 3), 512, 32, None),   # config 4 tile size / pattern
    ("fe27-nb768", lambda: matgen.fe27((10, 9, 9), 3), 768, 32, None),  # config 5 tile size / pattern
    ("box11-nb100-pw48", lambda: matgen.nd_like((11, 10, 10), 3), 100, 16, 48),  # ragged panels
    ("p3d15-nb192-pw64", lambda: matgen.poisson3d(15), 192, 24, 64),
]


@pytest.mark.parametrize("small", [False, True])
@pytest.mark.parametrize("name,gen,nb,nemin,pw", CASES)
def test_factor_matches_oracle(name, gen, nb, nemin, pw, small):
    """small=True: the analyse prunes the tree for four workers (spllt_prune_tree, small(:)) and the
    ORACLE takes its pruned-subtree path -- one task per small subtree with a private generated
    element and one extend-add at its root (reference factorization_mod:39-261, kernels_mod:97-821)
    -- so that this path of the oracle, too, is checked against the GPU engine (which factors the
    same L through its level-batched launches either way)."""
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw, prune=small, ncpu=4 if small else 1)
    f.factor(val).wait()
    got = f.get_factor()
    o, rc = oracle_factor(f, val, variant="mkl" if f.n > 4000 else "plain", nthreads=4, use_small=small)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) <= TOL_L
    assert np.all(got[~mask] == 0.0)
    b = A @ np.ones(f.n)
    x = f.solve(b)
    assert bwd_err(A, x, b) <= 1e-14
    np.testing.assert_allclose(x, np.ones(f.n), rtol=0, atol=1e-9)
    f.close()


# BASELINE configs 3/4/5 at their tile sizes with supernodes that really span several block
# columns of full width and several full-height tiles (reference kernels_mod:1261-1292,
# :2108-2237: update_block / update_between across block columns, K-segment walk).
BIG_CASES = [
    ("p3d40-nb384", lambda: matgen.poisson3d(40), 384),          # config 3 tile size
    ("fe27-16-nb512", lambda: matgen.fe27((16, 16, 16), 3), 512),  # config 4 tile size / pattern
    ("fe27-18-nb768", lambda: matgen.fe27((18, 18, 18), 3), 768),  # config 5 tile size / pattern
]


def _assert_multicolumn(f, nb):
    """the case must not degenerate: some block column is taller than two tiles
    and some supernode has at least three block columns"""
    assert int(f.sym("bcol_nrow").max()) > 2 * nb
    assert int(np.diff(f.sym("node_bcol0")).max()) >= 3
    assert int(f.sym("bcol_width").max()) == nb


@pytest.mark.parametrize("name,gen,nb", BIG_CASES)
def test_config_tile_sizes_multicolumn_nodes(name, gen, nb):
    A = gen()
    f, val = make_case(A, nb=nb, nemin=32)
    _assert_multicolumn(f, nb)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val, variant="mkl", nthreads=8)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) <= TOL_L
    assert np.all(got[~mask] == 0.0)
    b = A @ np.ones(f.n)
    x = f.solve(b)
    assert bwd_err(A, x, b) <= 1e-14
    f.close()


def test_kat_simple_c_through_spllt_all():
    """example/C/simple.c:25-75 call sequence via the one-shot entry point."""
    lib = api._lib.load()
    ptr = np.array([1, 3, 5, 6], dtype=np.int32)
    row = np.array([1, 2, 2, 3, 3], dtype=np.int32)
    val = np.array([2.0, -1.0, 2.0, -1.0, 2.0])
    rhs = np.ones(3)
    x = np.zeros(3)
    ak, fk = C.c_void_p(None), C.c_void_p(None)
    opt = api.spllt_options_t.default()
    info = api.spllt_inform_t()
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib.spllt_all(C.byref(ak), C.byref(fk), C.byref(opt), 3, 5, 1, 4, ptr.ctypes.data_as(ip),
                  row.ctypes.data_as(ip), val.ctypes.data_as(dp), x.ctypes.data_as(dp),
                  rhs.ctypes.data_as(dp), C.byref(info))
    assert info.flag == 0
    np.testing.assert_allclose(x, [1.5, 2.0, 1.5], rtol=1e-15)
    st = C.c_int()
    lib.spllt_deallocate_fkeep(C.byref(fk), C.byref(st))
    lib.spllt_deallocate_akeep(C.byref(ak), C.byref(st))
    assert fk.value is None and ak.value is None


def test_refactorize_same_pattern_and_device_val():
    torch = _torch()
    A = matgen.poisson2d(24)
    f, val = make_case(A, nb=16, nemin=8)
    f.factor(val).wait()
    L1 = f.get_factor()
    dval = torch.tensor(val * 2.0, device="cuda")
    torch.cuda.synchronize()
    f.factor_dev(dval.data_ptr()).wait()
    L2 = f.get_factor()
    np.testing.assert_allclose(L2, L1 * np.sqrt(2.0), rtol=1e-13, atol=1e-14)
    # ... and each of the two against the oracle's factorization of the same values
    mask = lower_mask(f)
    for got, v in ((L1, val), (L2, val * 2.0)):
        o, rc = oracle_factor(f, v)
        assert rc == 0
        assert rel_err(got, o.arena(), mask) <= TOL_L


def test_init_lfact_twin():
    """spllt_init_lfact_hip, the stream-taking twin of spllt_init_node_c / spllt_init_blk_c
    (reference kernels_mod:2301-2364, :2392-2423: lcol(map(1,i)) = val(map(2,i)) on a zeroed
    lcol) for the whole arena, against the oracle's own lmap (re-derived from the symbolic
    structure by the oracle, oracle/spllt_oracle.c): bit-exact."""
    torch = _torch()
    from oracle import pyoracle
    from spllt_amd import _lib
    A = matgen.nd_like((8, 8, 7), 2)
    f, val = make_case(A, nb=48, nemin=8)
    val = val * (1.0 + 0.001 * np.arange(val.size))            # every entry different
    o = pyoracle.OracleFactor.from_factorization(f)
    bc_off = np.asarray(f.sym("bcol_off"))
    arena = int(f.sym_info()["arena"])
    exp = np.zeros(arena)
    for b in range(bc_off.size):
        k = int(o.lib.spo_lmap_len(o.h, b))
        if k == 0:
            continue
        dst = np.ctypeslib.as_array(o.lib.spo_lmap_dst(o.h, b), shape=(k,))
        src = np.ctypeslib.as_array(o.lib.spo_lmap_src(o.h, b), shape=(k,))
        exp[bc_off[b] + dst] = val[src]
    assert np.count_nonzero(exp) == val.size
    dL = torch.zeros(arena, dtype=torch.float64, device="cuda")
    dv = _dev(torch, val)
    dd, ds = _dev(torch, np.asarray(f.sym("map_dst"), dtype=np.int64)), _dev(torch, np.asarray(f.sym("map_src"), dtype=np.int64))
    torch.cuda.synchronize()
    rc = _lib.load().spllt_init_lfact_hip(None, dL.data_ptr(), dv.data_ptr(), dd.data_ptr(), ds.data_ptr(), val.size)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(dL.cpu().numpy(), exp)


def test_not_positive_definite_is_reported():
    A = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(40, 40)).tolil()
    A[17, 17] = -3.0
    f, val = make_case(A.tocsc(), nb=8, nemin=4)
    f.factor(val)
    with pytest.raises(api.SplltError) as e:
        f.wait()
    assert e.value.flag == -20


def test_void_wait_leaves_the_flag_on_the_handle():
    """spllt_wait(void) cannot return what it found (reference src/spllt_mod.F90:172-182 is a
    bare taskwait): the flag stays on the handle -- spllt_hip_last_flag, and the next call that
    takes `info`."""
    A = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(40, 40)).tolil()
    A[17, 17] = -3.0
    f, val = make_case(A.tocsc(), nb=8, nemin=4)
    f.factor(val)                      # submits only
    f.lib.spllt_wait()                 # the reference's completion barrier: no handle, no result
    assert f.lib.spllt_hip_last_flag(f.fkeep) == -20
    with pytest.raises(api.SplltError) as e:
        f.solve(np.ones(f.n))
    assert e.value.flag == -20
    f.close()


def test_linearity_of_scaling():
    """Size-independent property: chol(c*A) = sqrt(c)*chol(A)."""
    A = matgen.nd_like((10, 9, 8), 2)
    f, val = make_case(A, nb=128, nemin=32)
    L1 = f.factor(val).wait().get_factor()
    L2 = f.factor(val * 4.0).wait().get_factor()
    np.testing.assert_allclose(L2, 2.0 * L1, rtol=1e-13, atol=1e-14)
    o, rc = oracle_factor(f, val * 4.0)      # (and not only with itself)
    assert rc == 0 and rel_err(L2, o.arena(), lower_mask(f)) <= TOL_L


# ---- per-kernel operator twins against the oracle's kernels -----------------
def _dev(torch, a, dtype=None):
    return torch.tensor(np.ascontiguousarray(a), device="cuda", dtype=dtype)


@pytest.mark.parametrize("m,n", [(5, 5), (64, 64), (100, 37), (256, 256), (300, 130), (384, 384)])
def test_factor_diag_block_twin(m, n):
    torch = _torch()
    from oracle import pyoracle
    olib = pyoracle.load("plain")
    rng = np.random.default_rng(m * 1000 + n)
    B = rng.standard_normal((n, n))
    S = B @ B.T + n * np.eye(n)
    tile = np.zeros((m, n))
    tile[:n] = np.tril(S)
    tile[n:] = rng.standard_normal((m - n, n))
    exp = tile.copy()
    assert olib.spo_factor_diag_block(m, n, exp.ctypes.data_as(C.POINTER(C.c_double))) == 0
    d = _dev(torch, tile)
    lib = api._lib.load()
    assert lib.spllt_factor_diag_block_hip(None, m, n, d.data_ptr(), None) == 0
    got = d.cpu().numpy()
    mask = np.ones((m, n), dtype=bool)
    mask[:n] = np.tril(np.ones((n, n), dtype=bool))
    assert rel_err(got, exp, mask) <= TOL_L
    assert np.all(got[~mask] == 0)


@pytest.mark.parametrize("m,n", [(7, 5), (64, 64), (200, 100), (256, 256), (129, 300)])
def test_solve_block_twin(m, n):
    torch = _torch()
    from oracle import pyoracle
    olib = pyoracle.load("plain")
    rng = np.random.default_rng(m + 7 * n)
    Lkk = np.tril(rng.standard_normal((n, n))) + n * np.eye(n)
    X = rng.standard_normal((m, n))
    exp = X.copy()
    dpp = C.POINTER(C.c_double)
    olib.spo_solve_block(m, n, exp.ctypes.data_as(dpp), np.ascontiguousarray(Lkk).ctypes.data_as(dpp))
    dk, dx = _dev(torch, Lkk), _dev(torch, X)
    assert api._lib.load().spllt_solve_block_hip(None, m, n, dk.data_ptr(), dx.data_ptr()) == 0
    assert rel_err(dx.cpu().numpy(), exp) <= TOL_L


@pytest.mark.parametrize("m,n,n1,diag", [(16, 16, 4, 1), (64, 64, 64, 0), (256, 256, 256, 1),
                                          (300, 200, 77, 0), (260, 256, 100, 1), (33, 17, 5, 0),
                                          # N = 128 + 11, 64 + 26, 64 + 3: the narrow 64-tiles (one column
                                          # fragment per wave) beside full ones, with and without a diagonal
                                          (200, 139, 50, 0), (300, 139, 256, 1), (200, 90, 33, 1), (131, 67, 20, 0)])
def test_update_block_twin(m, n, n1, diag):
    torch = _torch()
    from oracle import pyoracle
    olib = pyoracle.load("plain")
    rng = np.random.default_rng(m + n + n1)
    dest = rng.standard_normal((m, n))
    if diag:
        dest[:n] = np.tril(dest[:n])
    src1 = rng.standard_normal((n, n1))   # asymmetric operands: catches a swapped C map
    src2 = rng.standard_normal((m, n1)) if not diag else np.vstack([src1, rng.standard_normal((m - n, n1))])
    exp = dest.copy()
    dpp = C.POINTER(C.c_double)
    olib.spo_update_block(m, n, exp.ctypes.data_as(dpp), diag, n1,
                          np.ascontiguousarray(src1).ctypes.data_as(dpp),
                          np.ascontiguousarray(src2).ctypes.data_as(dpp))
    dd, d1, d2 = _dev(torch, dest), _dev(torch, src1), _dev(torch, src2)
    assert api._lib.load().spllt_update_block_hip(None, m, n, dd.data_ptr(), diag, n1,
                                                  d1.data_ptr(), d2.data_ptr()) == 0
    got = dd.cpu().numpy()
    mask = np.ones((m, n), dtype=bool)
    if diag:
        mask[:n] = np.tril(np.ones((n, n), dtype=bool))
    assert rel_err(got, exp, mask) <= TOL_L
    if diag:
        assert np.all(got[~mask] == 0)


@pytest.mark.parametrize("blkm,blkn,rls,cls,n1,diag", [(64, 48, 20, 11, 32, 0), (256, 256, 130, 90, 256, 0),
                                                        (128, 128, 70, 70, 40, 1), (256, 200, 150, 120, 9, 1),
                                                        (32, 32, 1, 1, 3, 0), (256, 256, 200, 139, 64, 0),
                                                        (256, 256, 180, 75, 100, 1)])
def test_update_between_and_expand_buffer_twins(blkm, blkn, rls, cls, n1, diag):
    """fused update_between (GEMM + scatter epilogue) and the stand-alone
    expand_buffer against spo_update_between pieces of the oracle."""
    torch = _torch()
    from oracle import pyoracle
    olib = pyoracle.load("plain")
    rng = np.random.default_rng(blkm + rls + cls + n1)
    row_list = np.sort(rng.choice(blkm, rls, replace=False)).astype(np.int32)
    col_list = np.sort(rng.choice(blkn, cls, replace=False)).astype(np.int32)
    if diag:
        rls = max(rls, cls)
        col_list = np.sort(rng.choice(min(blkm, blkn), cls, replace=False)).astype(np.int32)
        extra = np.setdiff1d(np.arange(blkm), col_list)[:rls - cls]
        row_list = np.concatenate([col_list, np.sort(extra) + 0]).astype(np.int32)
        row_list[cls:] = np.sort(row_list[cls:])
    csrc = rng.standard_normal((cls, n1))
    rsrc = rng.standard_normal((rls, n1))
    if diag:
        rsrc[:cls] = csrc
    ndiag = cls if diag else 0
    dest = rng.standard_normal((blkm, blkn))
    # oracle: buffer = -rsrc csrc^T then expand
    buf = -(rsrc @ csrc.T)
    exp = dest.copy()
    dpp, ipp = C.POINTER(C.c_double), C.POINTER(C.c_int)
    olib.spo_expand_buffer(exp.ctypes.data_as(dpp), blkn, row_list.ctypes.data_as(ipp), rls,
                           col_list.ctypes.data_as(ipp), cls, ndiag,
                           np.ascontiguousarray(buf).ctypes.data_as(dpp))
    lib = api._lib.load()
    dd, dc, dr = _dev(torch, dest), _dev(torch, csrc), _dev(torch, rsrc)
    drl, dcl = _dev(torch, row_list), _dev(torch, col_list)
    assert lib.spllt_update_between_hip(None, dd.data_ptr(), blkn, n1, dc.data_ptr(), cls,
                                        dr.data_ptr(), rls, drl.data_ptr(), dcl.data_ptr(), ndiag) == 0
    assert rel_err(dd.cpu().numpy(), exp) <= TOL_L
    d2, dbuf = _dev(torch, dest), _dev(torch, buf)
    assert lib.spllt_expand_buffer_hip(None, d2.data_ptr(), blkn, drl.data_ptr(), rls, dcl.data_ptr(),
                                       cls, ndiag, dbuf.data_ptr()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d2.cpu().numpy(), exp)  # pure adds of identical operands: bit-exact


def test_scatter_block_twin():
    torch = _torch()
    from oracle import pyoracle
    olib = pyoracle.load("plain")
    rng = np.random.default_rng(5)
    d_m, d_n, s_m, s_n = 90, 70, 40, 25
    rdest = np.sort(rng.choice(1000, d_m, replace=False)).astype(np.int32)
    cdest = np.sort(rng.choice(1000, d_n, replace=False)).astype(np.int32)
    rsrc = np.sort(rng.choice(rdest, s_m, replace=False)).astype(np.int32)
    csrc = np.sort(rng.choice(cdest, s_n, replace=False)).astype(np.int32)
    lds = 33
    src = rng.standard_normal((s_m, lds))
    dest = rng.standard_normal((d_m, d_n))
    exp = dest.copy()
    dpp, ipp = C.POINTER(C.c_double), C.POINTER(C.c_int)
    olib.spo_scatter_block(s_m, s_n, rsrc.ctypes.data_as(ipp), csrc.ctypes.data_as(ipp),
                           src.ctypes.data_as(dpp), lds, rdest.ctypes.data_as(ipp),
                           cdest.ctypes.data_as(ipp), exp.ctypes.data_as(dpp), d_n)
    t = [_dev(torch, a) for a in (rsrc, csrc, src, rdest, cdest, dest)]
    assert api._lib.load().spllt_scatter_block_hip(None, s_m, s_n, t[0].data_ptr(), t[1].data_ptr(),
                                                   t[2].data_ptr(), lds, t[3].data_ptr(), d_m,
                                                   t[4].data_ptr(), d_n, t[5].data_ptr(), d_n) == 0
    torch.cuda.synchronize()
    assert np.array_equal(t[5].cpu().numpy(), exp)  # integer index work + one subtract: bit-exact


def test_fortran_api_kat():
    """example/C/simple.c's case through the Fortran API module
    (spllt_amd/fortran/spllt_hip_mod.F90, built by __graft_entry__.build() with flang)."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spllt_amd",
                       "fortran", "kat_simple")
    if not os.path.exists(exe):
        pytest.skip("flang was not available at build time")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1.50000000  2.00000000  1.50000000" in r.stdout


def test_c_caller_inside_omp_parallel_single():
    """The C-ABI in its documented calling context (reference example/C/simple.c:52-75):
    analyse / factor / wait / solve / chkerr issued from inside `#pragma omp parallel` +
    `single` by whichever thread won the single, a second factorization submitted by
    another thread of the team (spllt_amd/c/omp_caller.c, built by __graft_entry__.build())."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spllt_amd", "c",
                       "omp_caller")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    for threads, g in (("4", "40"), ("7", "25")):
        r = subprocess.run([exe, g], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, OMP_NUM_THREADS=threads))
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"team={threads}" in r.stdout and "fail=0" in r.stdout, r.stdout


DIST_TOP = {"replicated": 16384, "distributed": 8192}    # engine flag bits 14 / 13


@pytest.mark.parametrize("top", ["replicated", "distributed"])
@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_engine_two_ranks_on_one_gpu(world, top):
    """Multi-GPU engine path on the one-GPU box: `world` rank-engines run in
    this process on the same device; the collectives of the exchanges are done with
    torch ops (what RCCL does across devices).  Top tree replicated (one all-reduce) or
    distributed over the ranks (reduce-scatter to the owners, a broadcast per finished
    block-column step, owner computes).  Every rank must end
    with its own subtrees + the whole top tree equal to the oracle's L."""
    _run_partitioned(matgen.nd_like((10, 9, 8), 2), 32, 8, world, 16, "plain", top=top)


@pytest.mark.parametrize("extra", [2, 4096, 512, 64], ids=["single-stream", "deterministic", "unfused", "no-slices"])
@pytest.mark.parametrize("top", ["replicated", "distributed"])
def test_partitioned_engine_variants(top, extra):
    """the partitioned program in the engine's other variants (single stream, deterministic
    assembly, no fused panel launches, no early slices), top tree replicated and distributed"""
    _run_partitioned(matgen.nd_like((12, 11, 10), 2), 48, 8, 3, 16, "plain", top=top, extra_flags=extra)


@pytest.mark.parametrize("top", ["replicated", "distributed"])
def test_partitioned_engine_eight_ranks_on_one_gpu(top):
    """the width the 8-GPU node runs: eight rank-engines on this device, deep top tree"""
    _run_partitioned(matgen.nd_like((16, 15, 14), 2), 64, 16, 8, None, "mkl", top=top)


@pytest.mark.parametrize("top", ["replicated", "distributed"])
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name,gen,nb", BIG_CASES)
def test_config_tile_sizes_partitioned(name, gen, nb, world, top):
    """configs 3/4/5 tile sizes through the 2- and 4-rank partition on one device"""
    _run_partitioned(gen(), nb, 32, world, None, "mkl", check_multicolumn=nb, top=top)


def _run_partitioned(A, nb, nemin, world, pw, variant, check_multicolumn=None, top="replicated", extra_flags=0):
    torch = _torch()
    fs, bufs = [], []
    for r in range(world):
        f, val = make_case(A, nb=nb, nemin=nemin, prune=True, ncpu=world, panel_width=pw,
                           engine_flags=DIST_TOP[top] | extra_flags)
        if check_multicolumn:
            _assert_multicolumn(f, check_multicolumn)
        xel = f.set_partition(r, world)
        assert xel > 0
        xb = torch.zeros(xel, dtype=torch.float64, device="cuda")
        f.set_exchange_buffer(xb.data_ptr())
        fs.append(f)
        bufs.append(xb)
    dval = torch.tensor(val, device="cuda")
    torch.cuda.synchronize()
    for f in fs:
        f.factor_dev(dval.data_ptr())
    nx = drive_exchanges(fs, bufs)
    for f in fs:
        f.wait()
    if top == "replicated":
        assert nx == 1
    else:
        kinds = fs[0].program("exchanges")[:, 0].tolist()
        nred = kinds.count(1)       # one reduce-scatter per level of the top tree, lowest first
        assert nred >= 1 and kinds[:nred] == [1] * nred and kinds[-1] == 3 and kinds.count(2) == nx - nred - 1 >= 1, kinds
        towner = fs[0].partition("top_bcol_owner")
        assert set(towner[towner >= 0].tolist()) == set(range(world))   # every rank owns part of the top tree
    # a rank's device arena holds only its own branches and the top tree, packed
    held = np.array([f.partition("arena_elems") for f in fs])
    top_elems = sum(int(fs[0].sym("bcol_nrow")[b]) * int(fs[0].sym("bcol_width")[b]) for b in fs[0].partition("top_bcols"))
    assert np.all(held[:, 1] == held[0, 1]) and np.all(held[:, 0] < held[:, 1])
    assert held[:, 0].sum() == held[0, 1] + (world - 1) * top_elems
    o, rc = oracle_factor(fs[0], val, variant=variant, nthreads=8)
    assert rc == 0
    ref = o.arena()
    mask = lower_mask(fs[0])
    owner = fs[0].partition("owner")
    bc_node = fs[0].sym("bcol_node")
    off, w, nr = fs[0].sym("bcol_off"), fs[0].sym("bcol_width"), fs[0].sym("bcol_nrow")
    for r, f in enumerate(fs):
        got = f.get_factor()
        mine = np.zeros_like(mask)
        for b in range(len(off)):
            if owner[bc_node[b]] in (r, -1):
                mine[off[b]:off[b] + nr[b] * w[b]] = True
        assert rel_err(got, ref, mask & mine) <= TOL_L
        assert np.all(got[~mine] == 0.0)
    # partitioned solve (spllt_hip_solve_dev phases 0/1/2, sums = the two all-reduces)
    n = fs[0].n
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, 3))
    B = A @ X
    pos = fs[0].sym("order")          # 0-based pivot position of variable i
    sptr = fs[0].sym("sptr")
    own = owner[np.repeat(np.arange(len(sptr) - 1), np.diff(sptr))]
    ys, masks = [], []
    for r in range(world):
        m = torch.tensor((own == r) | ((own < 0) & (r == 0)), device="cuda")
        Y = np.zeros((3, n))
        Y[:, pos] = B.T
        y = torch.tensor(Y, device="cuda") * m
        ys.append(y)
        masks.append(m)
    assert int(sum(m.sum().item() for m in masks)) == n      # every entry has exactly one owner
    torch.cuda.synchronize()
    for f, y in zip(fs, ys):
        f.solve_dev(y.data_ptr(), 3, 0, 0)
    total = torch.stack(ys).sum(dim=0)
    for y in ys:
        y.copy_(total)
    torch.cuda.synchronize()   # torch's stream is not the engines' stream
    for f, y, m in zip(fs, ys, masks):
        f.solve_dev(y.data_ptr(), 3, 0, 1)
        f.solve_dev(y.data_ptr(), 3, 0, 2)
        y *= m
    got = torch.stack(ys).sum(dim=0).cpu().numpy()[:, pos].T
    for q in range(3):
        assert bwd_err(A, got[:, q], B[:, q]) <= 1e-14
    np.testing.assert_allclose(got, X, rtol=0, atol=1e-9)
    with pytest.raises(api.SplltError) as ei:      # spllt_solve needs the caller's exchange
        fs[0].solve(B[:, 0])
    assert ei.value.flag == -98
    for f in fs:
        f.close()


@pytest.mark.parametrize("top", ["replicated", "distributed"])
def test_partitioned_not_posdef_is_reported_on_every_rank(top):
    """A non-positive pivot inside ONE rank's subtree: the indicator travels with the
    exchange buffer (replicated top tree: with the extend-add; distributed: in an exchange of
    its own at the end), so every rank-engine reports SPLLT_ERROR_NOT_POSDEF (-20) after the
    top tree instead of factorizing a garbage top tree and returning success."""
    torch = _torch()
    A = matgen.poisson2d(24).tolil()
    world = 2
    fs, bufs = [], []
    for r in range(world):
        f, val = make_case(A.tocsc(), nb=16, nemin=8, prune=True, ncpu=world, engine_flags=DIST_TOP[top])
        xb = torch.zeros(f.set_partition(r, world), dtype=torch.float64, device="cuda")
        f.set_exchange_buffer(xb.data_ptr())
        fs.append(f)
        bufs.append(xb)
    # break a diagonal entry that lives in a subtree owned by rank 1
    owner, sptr, order = fs[0].partition("owner"), fs[0].sym("sptr"), fs[0].sym("order")
    node_of_pos = np.repeat(np.arange(len(sptr) - 1), np.diff(sptr))
    var = int(np.nonzero(owner[node_of_pos[order]] == 1)[0][0])
    A[var, var] = -5.0
    n, ptr, row, val = api.csc_lower_1based(A.tocsc())
    dval = torch.tensor(val, device="cuda")
    torch.cuda.synchronize()
    for f in fs:
        f.factor_dev(dval.data_ptr())
        f.wait()          # phase 1 only drains the streams: no error yet
    seen = []

    def indicator(k, kind, src):
        if kind in (0, 3):     # the exchanges that carry the indicator (last element they cover)
            e = int(fs[0].program("exchanges")[k][3])
            seen.append(sum(float(s[e - 1].item()) for s in src))
    drive_exchanges(fs, bufs, on_exchange=indicator)
    # replicated: exactly one rank raised the indicator at the exchange point; distributed: the
    # indicator is exchanged at the very end, by when the broken values have reached the top
    # tree, where other ranks' pivots may fail as well
    assert len(seen) == 1 and (seen[0] == 1.0 if top == "replicated" else seen[0] >= 1.0)
    for r, f in enumerate(fs):
        with pytest.raises(api.SplltError) as e:
            f.wait()
        assert e.value.flag == -20, r
    for f in fs:
        f.close()


@pytest.mark.parametrize("gen,nb", [(lambda: matgen.nd_like((14, 13, 12), 2), 64),
                                    (lambda: matgen.poisson3d(22), 48),
                                    (lambda: matgen.fe27((8, 7, 7), 3), 256)])
def test_deterministic_engine_is_bit_reproducible(gen, nb):
    """Engine flag 4096: no atomic adds (inter-node updates through a buffer + ordered
    gather, k_gather).  Two factorizations of the same values must give bit-identical
    factors (np.array_equal) -- the reference's OpenMP path serialises the updates of a
    destination (task_mod:1239-1241) and has this property; the default engine (fp64
    atomics in the scatter epilogue) only reproduces L to rounding.  Also checked against
    the oracle, and re-factorization with other values in between."""
    A = gen()
    f, val = make_case(A, nb=nb, nemin=16, engine_flags=4096)
    assert (f.program("launches")[:, 0] == 6).any()
    L1 = f.factor(val).wait().get_factor()
    f.factor(val * 3.0).wait()
    L2 = f.factor(val).wait().get_factor()
    assert np.array_equal(L1, L2)
    g, _ = make_case(A, nb=nb, nemin=16, engine_flags=4096)      # a second engine instance
    assert np.array_equal(g.factor(val).wait().get_factor(), L1)
    o, rc = oracle_factor(f, val, variant="mkl" if f.n > 4000 else "plain", nthreads=4)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(L1, o.arena(), mask) <= TOL_L
    assert np.all(L1[~mask] == 0.0)
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14
    f.close()
    g.close()


def test_golden_vectors_gpu():
    """tests/golden/dense_chol_*.npz (dense LAPACK factors for a fixed order):
    the HIP path must reproduce them to 1e-12 and solve to x = 1."""
    import glob
    import os
    from test_golden import expected_arena
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dense_chol_*.npz")))
    assert paths
    for path in paths:
        g = np.load(path)
        f = api.Factorization(int(g["n"]), g["ptr"], g["row"], nb=16, nemin=4, prune_tree=False,
                              order=g["order_in"], panel_width=16)
        exp, _ = expected_arena(f, g)
        got = f.factor(g["val"]).wait().get_factor()
        mask = lower_mask(f)
        assert np.abs(got - exp)[mask].max() <= TOL_L * np.abs(exp).max()
        np.testing.assert_allclose(f.solve(g["b"]), g["x"], rtol=0, atol=1e-11)


@pytest.mark.parametrize("chain4", [1, 0])
@pytest.mark.parametrize("flags", [0, 2, 512, 4096])
@pytest.mark.parametrize("gen,nb,pw", [(lambda: matgen.nd_like((12, 11, 10), 2), 160, 32),     # 5 panels: 4 + 1
                                       (lambda: matgen.nd_like((14, 13, 12), 3), 256, None),    # 256 = 4 x 64
                                       (lambda: matgen.nd_like((14, 13, 12), 3), 200, None),    # 200 = 3 x 64 + 8
                                       (lambda: matgen.fe27((9, 8, 8), 3), 200, 48),            # 4 x 48 + 8
                                       (lambda: matgen.fe27((10, 9, 9), 3), 384, None)])        # 256 + 128
def test_chain_blocks_match_oracle(chain4, flags, gen, nb, pw, monkeypatch):
    """Chain blocks of up to four panels (default; SPLLT_CHAIN4=0: one panel per chain step):
    k_chain_block factors the whole diagonal block of a chain block in one workgroup and emits the
    panels' inverses, k_trsm_rows solves the rows below against it -- two dependent launches per
    4 pw columns of the panel chain instead of twelve.  Block columns of 2, 3, 4 and 5+ panels,
    ragged last panels, wider than one chain block (the right-looking update of the rest of the
    block column in between); the device solve reads the same per-panel inverses either way."""
    monkeypatch.setenv("SPLLT_CHAIN4", str(chain4))
    A = gen()
    f, val = make_case(A, nb=nb, nemin=16, panel_width=pw, engine_flags=flags)
    kinds = f.program("launches")[:, 0]
    assert ((kinds == 8).any() and (kinds == 9).any()) == bool(chain4)
    assert max(f.sym("bcol_width")) > (pw or 64), "the case is meant to have block columns of several panels"
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14


@pytest.mark.parametrize("flags", [2, 64, 66, 256, 512, 1024, 2048, 4096, 4098])
@pytest.mark.parametrize("cb", [None, 32, 96, 160])
def test_engine_variants_match_oracle(flags, cb):
    """single-stream program (2), inter-node updates only at the end of a level (64), both
    (66), no CU reservation (256), no fused panel launches (512; they are used when the chain
    block is one panel, cb = 32 here), zone pipeline forced on / off (1024 / 2048), deterministic
    (4096); chain block = one panel (default), several panels per sub-tile, whole block columns."""
    A = matgen.nd_like((12, 11, 10), 2)
    f, val = make_case(A, nb=160, nemin=16, panel_width=32, engine_flags=flags, chain_block=cb)
    assert f.program("chain_block") == 32          # always one panel (the knob is ignored)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14


@pytest.mark.parametrize("graph", [0, 1, 2])
@pytest.mark.parametrize("budget", [40, 300, 100000])
@pytest.mark.parametrize("gen,nb,pw,nemin", [(lambda: matgen.poisson2d(64), 64, None, 8),        # deep subtrees of tiny nodes
                                             (lambda: matgen.poisson2d(12), 64, None, 4),        # the whole tree in one task
                                             (lambda: matgen.nd_like((12, 11, 10), 2), 64, None, 16),
                                             (lambda: matgen.fe27((8, 7, 6), 3), 48, 24, 8),     # ragged panels of 24
                                             (lambda: matgen.poisson3d(14), 256, None, 32)])
def test_subtree_tasks_match_oracle(gen, nb, pw, nemin, budget, graph, monkeypatch):
    """k_subtree (L_SUBTREE; opt-in): a whole small subtree per workgroup -- Cholesky, solve and update
    units of its nodes in post-order, what leaves the subtree through the generated-element scratch,
    one extend-add from the root (the reference's a20-a25: src/spllt_factorization_mod.F90:39-261).
    Against the oracle; over new values on the same pattern (the scratch is zero again after every
    factorization -- also after one that failed); eager and as a graph node."""
    monkeypatch.setenv("SPLLT_SUBTREES", "1")
    monkeypatch.setenv("SPLLT_SUBTREE_US", str(budget))
    monkeypatch.setenv("SPLLT_HIP_GRAPH", str(graph))
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin, panel_width=pw)
    L = f.program("launches")
    assert (L[:, 0] == 10).sum() == 1 and len(f.program("sub_tasks")) == L[0, 3] > 0
    for scale in (1.0, 2.5):
        got = f.factor(val * scale).wait().get_factor()
        o, rc = oracle_factor(f, val * scale)
        assert rc == 0
        assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    b = (A * 2.5) @ np.ones(f.n)
    assert bwd_err(A * 2.5, f.solve(b), b) <= 1e-14
    bad = val.copy()
    bad[0] = -1.0                                   # not positive definite: reported, and the next one is clean
    with pytest.raises(api.SplltError) as ei:
        f.factor(bad).wait()
    assert ei.value.flag == -20
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val)
    assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    f.close()


@pytest.mark.parametrize("gen,nb,nemin", [(lambda: matgen.poisson2d(128), 256, 32),          # BASELINE config 1
                                          (lambda: matgen.poisson2d(48), 32, 16),
                                          (lambda: matgen.nd_like((12, 11, 10), 2), 64, 16)])
def test_default_replay_of_small_factorizations(gen, nb, nemin, monkeypatch):
    """The library's default for a small factorization: ONE chain of kernel nodes over the SINGLE-STREAM
    program (the test-suite otherwise keeps the multi-stream program, conftest.py).  Against the oracle,
    over new values on the same pattern, and a not-positive-definite input reported through the graph."""
    monkeypatch.setenv("SPLLT_CHAIN_GRAPH_SERIAL", "1")
    monkeypatch.delenv("SPLLT_HIP_GRAPH", raising=False)
    A = gen()
    f, val = make_case(A, nb=nb, nemin=nemin)
    L = f.program("launches")
    assert (L[:, 6] == 0).all() and (L[:, 7:] == -1).all()
    for scale in (1.0, 0.25, 3.0):
        got = f.factor(val * scale).wait().get_factor()
        o, rc = oracle_factor(f, val * scale)
        assert rc == 0
        assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    b = (A * 3.0) @ np.ones(f.n)
    assert bwd_err(A * 3.0, f.solve(b), b) <= 1e-14
    bad = val.copy()
    bad[0] = -1.0
    with pytest.raises(api.SplltError) as ei:
        f.factor(bad).wait()
    assert ei.value.flag == -20
    f.close()


@pytest.mark.parametrize("flags", [32768, 65536, 65536 | 4096, 32768 | 512, 65536 | 2048])
def test_graph_replay_matches_oracle(flags):
    """SURVEY 8(f) row f1, analyse once / factorize many (reference kernels_mod:2301-2364): the
    factorization of a pattern as ONE HIP graph built from the program tables (flag bit 15: a
    chain of kernel nodes in program order; bit 16: the DAG of the multi-stream program), replayed
    for new values of the same pattern.  Every replay against the oracle."""
    A = matgen.nd_like((12, 11, 10), 2)
    f, val = make_case(A, nb=160, nemin=16, panel_width=32, engine_flags=flags)
    for scale in (1.0, 3.0, 0.5):          # same pattern, new values: the graph is replayed
        v = val * scale
        got = f.factor(v).wait().get_factor()
        o, rc = oracle_factor(f, v)
        assert rc == 0
        assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    b = (A * 0.5) @ np.ones(f.n)
    assert bwd_err(A * 0.5, f.solve(b), b) <= 1e-14
    # a matrix that is not positive definite is reported through the graph's own copy of the flag
    bad = val.copy()
    bad[0] = -1.0
    with pytest.raises(api.SplltError) as ei:
        f.factor(bad).wait()
    assert ei.value.flag == -20
    f.close()


@pytest.mark.parametrize("gen,nb", [(lambda: matgen.poisson3d(40), 384), (lambda: matgen.fe27((20, 20, 18), 3), 256)])
def test_fused_panel_launches_larger_than_the_chip(gen, nb, monkeypatch):
    """k_panel: the diagonal block and the next pivot rows are read by every workgroup of a
    unit and overwritten by the step; the workgroup that read them LAST writes them.  With more
    workgroups in a launch than the chip holds at once (one per CU), late workgroups start after
    early ones have finished - they must still find the blocks unsolved.  The default only fuses
    launches of <= 64 workgroups; here every step is fused."""
    monkeypatch.setenv("SPLLT_FUSED_PANEL_MAX", "1000000")
    monkeypatch.setenv("SPLLT_SUBTREES", "0")      # (the leaves stay in the level-batched launches)
    A = gen()
    f, val = make_case(A, nb=nb, nemin=32)
    L = f.program("launches")
    assert not (L[:, 0] == 4).any() and (L[(L[:, 0] == 7), 3] > 256).any()
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val, variant="mkl", nthreads=8)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    got2 = f.factor(val).wait().get_factor()       # the counters were left at zero
    assert rel_err(got2, o.arena(), lower_mask(f)) <= TOL_L
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14


@pytest.mark.parametrize("nb,pw,cb,flags", [(48, 5, None, 128), (100, 40, 100, 130), (130, 48, 96, 192),
                                            (33, 12, 24, 128), (256, 64, 256, 128), (200, 24, 72, 4224),
                                            (130, 48, None, 4224), (256, 64, None, 1152)])
def test_no_kernel_reads_uninitialised_lds(nb, pw, cb, flags):
    """Engine flag 128: before EVERY kernel launch of the factorization a poison kernel
    fills the whole LDS of every CU with signalling-NaN bit patterns.  A kernel that reads
    LDS it has not written (zero padding assumed, columns past a ragged panel, ...) then
    produces NaNs deterministically instead of depending on what the previous kernel left
    there.  Ragged shapes: panel widths that are no multiple of 16 or 4 and do not divide
    the tile size, tile sizes that divide nothing.  Run once."""
    A = matgen.fe27((5, 4, 4), 3) if nb < 200 else matgen.nd_like((10, 9, 9), 3)
    f, val = make_case(A, nb=nb, nemin=4, panel_width=pw, engine_flags=flags, chain_block=cb)
    got = f.factor(val).wait().get_factor()
    assert np.isfinite(got).all()
    o, rc = oracle_factor(f, val)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(f)) <= TOL_L
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14


@pytest.mark.parametrize("gen,nb", [(lambda: matgen.poisson2d(40), 16), (lambda: matgen.nd_like((11, 10, 9), 2), 64),
                                    (lambda: matgen.poisson3d(14), 384), (lambda: matgen.fe27((7, 6, 6), 3), 768)])
def test_device_solve_jobs_and_multiple_rhs(gen, nb):
    """spllt_solve on the device-resident factor: job 0 = both sweeps, job 1
    then job 2 = the same, several right-hand sides (reference
    src/spllt_solve_mod.F90:203-221); checked against the oracle's solve and
    the reference's backward-error bar."""
    A = gen()
    f, val = make_case(A, nb=nb, nemin=16)
    f.factor(val).wait()
    rng = np.random.default_rng(0)
    X = rng.standard_normal((f.n, 7))     # 7 right-hand sides = sweeps of 4, 2 and 1
    B = A @ X
    got = f.solve(B)
    for r in range(7):
        assert bwd_err(A, got[:, r], B[:, r]) <= 1e-14
    np.testing.assert_allclose(f.solve(B[:, :3])[:, 2], got[:, 2], rtol=1e-12, atol=1e-12)
    y = f.solve(B[:, 0], job=1)
    x2 = f.solve(y, job=2)
    np.testing.assert_allclose(x2, got[:, 0], rtol=1e-12, atol=1e-12)
    o, rc = oracle_factor(f, val)
    np.testing.assert_allclose(got[:, 1], o.solve(B[:, 1]), rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("nb", [32, 64, 128])
@pytest.mark.parametrize("prune", [False, True])
def test_stress_harness_sweep_from_a_matrix_file(nb, prune, tmp_path):
    """The reference's stress harness (scripts/stress_test.sh:24-128 around test/test_solve_phasis.F90:
    a Rutherford-Boeing file read with rb_options%values = 3, nb in 32..128, with and without tree
    pruning, nrhs in 1..10, 16..128, and after every solve the scaled backward error against 1e-14
    per right-hand side) -- here with the file going through the LIBRARY's reader
    (spllt_hip_read_rb) and everything else through the C-ABI."""
    rng = np.random.default_rng(5)
    M = sp.random(600, 600, density=0.01, random_state=np.random.RandomState(9), format="csr")
    pat = ((M + M.T) + sp.eye(600)).tocsc()
    path = tmp_path / "stress.rb"
    matgen.write_rb(str(path), pat)
    n, ptr, row, val = matgen.read_file_c(str(path), "rb", values=3)       # diagonally dominant values
    A = sp.csc_matrix((val, row - 1, ptr - 1), shape=(n, n))
    A = (A + sp.tril(A, -1).T).tocsc()
    f = api.Factorization(n, ptr.astype(np.int32), row.astype(np.int32), nb=nb, nemin=8, prune_tree=prune, ncpu=3)
    f.factor(val).wait()
    ntest = 0
    for nrhs in (1, 2, 3, 5, 10, 16, 128):
        X = rng.standard_normal((n, nrhs))
        B = A @ X
        got = f.solve(B if nrhs > 1 else B[:, 0])
        got = got.reshape(n, nrhs)
        for r in range(nrhs):
            assert bwd_err(A, got[:, r], B[:, r]) <= 1e-14, (nrhs, r)
        ntest += 1
    # forward and backward sweeps as separate calls (job 1, then job 2), several right-hand sides
    B = A @ np.ones((n, 4))
    x = f.solve(f.solve(B, job=1), job=2)
    np.testing.assert_allclose(x, np.ones((n, 4)), rtol=0, atol=1e-10)
    assert ntest == 7
    f.close()


def test_bench_workload_full_size_properties():
    """BASELINE config 2 stand-in at its full size (n = 72 324, 755 GFLOP): the
    factor agrees with the CPU oracle (MKL build), the reference's residual bar
    holds (src/utils_mod.F90:462-467), a second factorization of the same values
    reproduces L up to the order of the atomic adds, and chol(4A) = 2 chol(A)."""
    A, order, cfg = matgen.build_config("nd24k_like", 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    L1 = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val, variant="mkl", nthreads=8)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(L1, o.arena(), mask) <= TOL_L
    b = A @ np.ones(n)
    x = f.solve(b)
    assert bwd_err(A, x, b) <= 1e-14
    L2 = f.factor(val).wait().get_factor()
    assert np.abs(L2 - L1).max() <= 1e-13 * np.abs(L1).max()
    L4 = f.factor(4.0 * val).wait().get_factor()
    assert np.abs(L4 - 2.0 * L1).max() <= 1e-13 * np.abs(L1).max()


@pytest.mark.timeout(1200)
@pytest.mark.skipif(not os.environ.get("SPLLT_TEST_FULL"),
                    reason="opt-in (SPLLT_TEST_FULL=1): 40-100 GB of host memory and 1-1.5 min of CPU oracle per case")
@pytest.mark.parametrize("name", ["poisson3d_128", "serena_like", "flan_like"])
def test_large_configs_full_size_against_the_oracle(name):
    """BASELINE configs 3, 5 and 4 at their FULL size on one GPU: every entry of L (1.6 / 2.1 / 4.1 G) against
    the CPU oracle (MKL + OpenMP tasks, all host cores), max|dL| / max|L| <= 1e-12, and the reference's
    residual bar.  12-48 TFLOP each: the oracle takes 15-60 s, the GPU 0.25-0.8 s.  Opt-in, and meant to be
    run by itself (`SPLLT_TEST_FULL=1 pytest tests/test_gpu_parity.py -m gpu -k full_size_against`: 58 + 37 +
    90 s on the box; inside the whole suite the oracle's factorization ran into the per-test timeout once)."""
    A, order, cfg = matgen.build_config(name, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=cfg["nb"], nemin=32, prune_tree=False, order=order)
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val, variant="mkl", nthreads=min(16, len(os.sched_getaffinity(0))))
    assert rc == 0
    ref = o.arena()
    # the strict upper triangle of the diagonal tiles is never read by anybody:
    # cleared on both sides, then the arenas are compared whole, in pieces
    off, bw = f.sym("bcol_off"), f.sym("bcol_width")
    tri = {}
    for b in range(len(off)):
        w = int(bw[b])
        if w not in tri:
            tri[w] = np.triu_indices(w, 1)
        for arr in (got, ref):
            arr[int(off[b]):int(off[b]) + w * w].reshape(w, w)[tri[w]] = 0.0
    err, top = 0.0, 0.0
    step = 1 << 27
    for a in range(0, got.size, step):
        err = max(err, float(np.abs(got[a:a + step] - ref[a:a + step]).max()))
        top = max(top, float(np.abs(ref[a:a + step]).max()))
    assert err <= TOL_L * top, (err, top)
    b = A @ np.ones(n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14
    f.close()


def test_solve_dev_on_device_vectors():
    """spllt_hip_solve_dev: the substitution on caller-owned device vectors in
    pivot order equals spllt_solve (5 right-hand sides, jobs 0 / 1 then 2)."""
    torch = _torch()
    A = matgen.nd_like((11, 10, 9), 2)
    f, val = make_case(A, nb=64, nemin=16)
    f.factor(val).wait()
    rng = np.random.default_rng(3)
    X = rng.standard_normal((f.n, 5))
    B = A @ X
    pos = f.sym("order")
    Y = np.zeros((5, f.n))
    Y[:, pos] = B.T
    y = torch.tensor(Y, device="cuda")
    f.solve_dev(y.data_ptr(), 5, 0, -1)
    got = y.cpu().numpy()[:, pos].T
    np.testing.assert_allclose(got, f.solve(B), rtol=1e-12, atol=1e-12)
    y2 = torch.tensor(Y, device="cuda")
    f.solve_dev(y2.data_ptr(), 5, 1, -1)
    f.solve_dev(y2.data_ptr(), 5, 2, -1)
    np.testing.assert_allclose(y2.cpu().numpy()[:, pos].T, got, rtol=1e-12, atol=1e-12)


def test_sync_watchdog_reports_where_the_program_stands():
    """Every blocking wait of the engine has a deadline (SPLLT_HIP_TIMEOUT_S, default 180 s): a
    device that does not finish makes spllt_wait FAIL (flag -30) with a report of the first
    launches whose events have not fired, instead of blocking the caller forever.  Exercised with
    a deadline the factorization cannot meet, in a process of its own (the deadline is read once)."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from spllt_amd import api, matgen\n"
        "from helpers import make_case\n"
        "f, val = make_case(matgen.nd_like((24, 22, 20), 3), nb=128, nemin=16)\n"
        "try:\n"
        "    f.factor(val).wait()\n"
        "    print('FINISHED')\n"
        "except api.SplltError as e:\n"
        "    print('FLAG', e.flag); print(str(e))\n"
        "import time; t0 = time.time(); f.close(); print('CLOSED in %%.2f s' %% (time.time() - t0))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, SPLLT_HIP_TIMEOUT_S="0.0000001"))
    out = r.stdout + r.stderr
    assert "FLAG -30" in out, out
    assert "did not drain" in out and "stream 0" in out, out
    # (the launches that have not finished are listed -- unless the device got through the small
    # factorization between the missed deadline and the report, which then says the streams are idle)
    assert "has not finished" in out or "stream 0 idle" in out, out
    # the handle whose wait ran into the deadline closes at once: its engine is poisoned, nothing
    # of it is synchronised, returned to the pools or freed (an unbounded hipStreamSynchronize in
    # the destructor would sit on a device that is really stuck)
    assert "CLOSED in" in out and r.returncode == 0, out


@pytest.mark.parametrize("gen,nb", [(lambda: matgen.poisson3d(40), 384), (lambda: matgen.nd_like((12, 11, 10), 2), 160)])
def test_host_program_equals_engine_program(gen, nb):
    """The host-only program (spllt_hip_program_get before the first factorization: what the CPU
    DAG-conflict test, the numpy emulator and the partition model look at) and the program the
    engine runs come from ONE options mapping (engine.cpp schedule_options), incl. the decisions
    taken from the problem (latency-bound or throughput-bound: zones, tile thresholds)."""
    A = gen()
    f, val = make_case(A, nb=nb, nemin=32)
    before = {k: np.array(f.program(k)) for k in ("launches", "tiles", "units")}
    f.factor(val).wait()
    for k, v in before.items():
        after = np.array(f.program(k))
        assert after.shape == v.shape and np.array_equal(after, v), k
    f.close()


def test_timeline_of_the_real_program():
    """spllt_hip_timeline: one factorization submitted exactly like spllt_factor's, on timing-enabled
    events of the call -- the completion time of every event the program records.  The times obey
    the program (events of one stream in order, nothing before what it waited for), and the
    factor the call leaves behind is the factor of the values it was given (against the oracle)."""
    A = matgen.nd_like((12, 11, 10), 2)
    f, val = make_case(A, nb=64, nemin=16)
    f.factor(val)                       # (still in flight: the call drains it first)
    t = f.timeline(val * 2.0)
    L = f.program("launches")
    assert len(t) == len(L) + 1
    rec = L[:, 7] >= 0
    assert rec.sum() > 10
    assert (t[:-1][rec] >= 0).all() and (t[:-1][~rec] == -1).all()
    assert t[:-1].max() - 1e-3 <= t[-1] < 1e3
    for st in set(L[:, 6].tolist()):
        tt = t[:-1][(L[:, 6] == st) & rec]
        assert (np.diff(tt) >= -2e-3).all(), st          # (event resolution ~1 us)
    at = {int(L[i, 7]): t[i] for i in range(len(L)) if rec[i]}
    for i in np.where(rec)[0]:
        for w in L[i, 8:12]:
            if w >= 0:
                assert t[i] >= at[int(w)] - 2e-3, (i, int(w))
    o, rc = oracle_factor(f, val * 2.0)
    assert rc == 0
    assert rel_err(f.get_factor(), o.arena(), lower_mask(f)) <= TOL_L
    f.close()


@pytest.mark.parametrize("kind", ["single-columns", "hand-amalgamated"])
@pytest.mark.parametrize("gen,nb", [(lambda: matgen.poisson2d(20), 8), (lambda: matgen.nd_like((9, 8, 8), 2), 64),
                                    (lambda: matgen.fe27((6, 5, 5), 3), 96)])
def test_foreign_symbolic_factorization_on_the_gpu(kind, gen, nb):
    """spllt_hip_analyse_symbolic takes the symbolic factorization the reference
    gets from SSIDS (src/spllt_analyse_mod.F90:129-158) -- here two partitions the product's own
    analyse would never produce: every column its own node, and a hand-amalgamated coarser one
    -- and spllt_factor runs on exactly that tree: the HIP path against the oracle on the same
    (foreign) symbolic structure."""
    from helpers import quintuple_hand_amalgamated, quintuple_single_columns
    A = gen()
    base, val = make_case(A, nb=nb, nemin=1)
    quint = quintuple_single_columns(base) if kind == "single-columns" else quintuple_hand_amalgamated(base)
    n, ptr, row, _ = api.csc_lower_1based(A)
    g = api.Factorization(n, ptr, row, nb=nb, nemin=32, symbolic=quint)
    assert g.sym_info()["ordering"] == "symbolic"
    if kind == "single-columns":
        assert g.sym_info()["nnodes"] == n
    else:
        assert g.sym_info()["nnodes"] < base.sym_info()["nnodes"]
    got = g.factor(val).wait().get_factor()
    o, rc = oracle_factor(g, val)
    assert rc == 0
    assert rel_err(got, o.arena(), lower_mask(g)) <= TOL_L
    b = A @ np.ones(n)
    assert bwd_err(A, g.solve(b), b) <= 1e-14
    g.close()
    base.close()


def test_submission_deadline():
    """spllt_factor itself is under a deadline: engine creation, the staging of val and the
    launches run on the library's helper thread; a runtime call that does not come back (simulated:
    SPLLT_HIP_TEST_STALL_MS) makes spllt_factor FAIL with flag -30 and the name of the step it
    sat in, every later call of the process fail at once, and the handle close without touching
    the engine that the stuck thread still holds."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, time; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from spllt_amd import api, matgen\n"
        "from helpers import make_case\n"
        "f, val = make_case(matgen.poisson2d(24), nb=32, nemin=8)\n"
        "g, val2 = make_case(matgen.poisson2d(16), nb=32, nemin=8)\n"
        "t0 = time.time()\n"
        "try:\n"
        "    f.factor(val).wait(); print('FINISHED')\n"
        "except api.SplltError as e:\n"
        "    print('FLAG', e.flag, 'after %%.1f s' %% (time.time() - t0)); print(str(e))\n"
        "try:\n"
        "    g.factor(val2).wait(); print('SECOND FINISHED')\n"
        "except api.SplltError as e:\n"
        "    print('SECOND FLAG', e.flag); print(str(e))\n"
        "f.close(); g.close(); print('CLOSED after %%.1f s' %% (time.time() - t0))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, SPLLT_HIP_SUBMIT_TIMEOUT_S="1", SPLLT_HIP_TEST_STALL_MS="60000"))
    t_child = time.time() - t0
    out = r.stdout + r.stderr
    # the process that detected the stall also EXITS: the atexit teardown of the pools touches nothing once
    # the runtime is marked wedged (the stalled call here takes a minute; before round 4 the handler ran
    # hipFree / hipHostFree behind it)
    assert t_child < 45.0, (t_child, out[-2000:])
    assert "FLAG -30" in out and "submission did not return within 1 s; last step: factor: H2D of val" in out, out
    assert "SECOND FLAG -30" in out and "did not return from an earlier call" in out, out
    assert "CLOSED after" in out, out
    t_closed = float(out.split("CLOSED after")[1].split()[0])
    assert t_closed < 4.0, out       # (nobody waited for the minute the stuck call takes)


def test_bench_contract_one_gpu():
    """`python bench.py` on a shrunken workload: ONE JSON line with the driver's fields, the
    roofline object (dominant kernel, alone and inside the program) and the CPU baseline of the
    oracle on the same workload; the accuracy gate inside must pass."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--scale", "0.5", "--no-extra-configs"], capture_output=True, text=True, timeout=280,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "GFLOP/s" and d["value"] > 0 and "workload" in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 78.6
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac_alone"] <= 1
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    chk = d["detail"]["check"]
    assert chk["bwd_err"] <= 1e-14 and chk["max_relerr_L_vs_cpu"] <= 1e-12


@pytest.mark.parametrize("flags", [0, 2, 4096])
@pytest.mark.parametrize("gen,nb,pw", [(lambda: matgen.fe27((14, 13, 12), 3), 200, None),     # K windows of 200 = 12 x 16 + 8
                                       (lambda: matgen.nd_like((16, 15, 14), 3), 150, 48),    # 150 = 9 x 16 + 6, panels of 48
                                       (lambda: matgen.poisson3d(30), 37 * 8, None),           # nb = 296
                                       (lambda: matgen.nd_like((13, 12, 11), 3), 111, 37)])    # odd widths: rows 8-byte aligned only
def test_dma_update_kernel_on_every_large_launch(gen, nb, pw, flags, monkeypatch):
    """k_update_dma128 (operand tiles DMA'd straight into swizzled LDS stages): by default only
    launches of >= 4096 tiles use the 128-tile; here every update with M, N >= 96 does, on shapes
    whose K windows are no multiple of the 16-column chunks (whole chunks are loaded, the excess is
    cleared in LDS) and whose rows are only 8-byte aligned, in the default, single-stream and
    deterministic (BUFFER epilogue) engines."""
    monkeypatch.setenv("SPLLT_TILE_SMALL", "0")
    monkeypatch.setenv("SPLLT_TILE_TINY", "0")
    monkeypatch.setenv("SPLLT_SUBTREES", "0")      # (the leaves' updates stay 128-tile launches)
    A = gen()
    f, val = make_case(A, nb=nb, nemin=16, panel_width=pw, engine_flags=flags)
    L = f.program("launches")
    assert ((L[:, 0] == 1) & (L[:, 4] == 128) & (L[:, 3] > 0)).sum() >= 5, "expected 128-tile launches"
    got = f.factor(val).wait().get_factor()
    o, rc = oracle_factor(f, val, variant="mkl", nthreads=8)
    assert rc == 0
    mask = lower_mask(f)
    assert rel_err(got, o.arena(), mask) <= TOL_L
    assert np.all(got[~mask] == 0.0)
    b = A @ np.ones(f.n)
    assert bwd_err(A, f.solve(b), b) <= 1e-14
