"""CPU test of the drop-in boundary: libspllt_hip.so loads without a GPU,
exports every symbol include/*.h declares, and the non-GPU entry points
(analyse, workspace sizing, task-manager tokens, chkerr, deallocate) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spllt_amd import _lib, api, matgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(spllt_\w+)\s*\(", txt)) - {"SPLLT_OPTIONS_NULL"}


def test_every_declared_symbol_is_exported():
    lib = _lib.load()
    names = _declared("spllt_iface.h") | _declared("spllt_hip.h")
    assert set(_lib.IFACE_SYMBOLS) <= names and set(_lib.HIP_SYMBOLS) <= names
    missing = [s for s in sorted(names) if not hasattr(lib, s)]
    assert not missing, missing


def test_struct_layouts_match_reference_abi():
    assert C.sizeof(api.spllt_options_t) == 14 * 4     # include/spllt_iface.h:14-31
    assert C.sizeof(api.spllt_inform_t) == 6 * 4       # :49-57
    o = api.spllt_options_t.default()
    assert (o.nb, o.nemin, o.prune_tree, o.ncpu, o.min_width_blas) == (16, 32, 1, 1, 8)


def test_analyse_without_gpu_and_factor_fails_loudly():
    A = matgen.poisson2d(10)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=8)
    assert f.info.flag == 0 and f.info.num_nodes == f.sym_info()["nnodes"]
    assert sorted(f.order[:n].tolist()) == list(range(1, n + 1))   # 1-based positions
    ws = C.c_long()
    f.lib.spllt_solve_workspace_size(f.fkeep, 2, 3, C.byref(ws))
    assert ws.value == n * 3 + (f.sym_info()["maxmn"] + n) * 3 * 2  # src/spllt_data_mod.F90:655
    import torch
    if not torch.cuda.is_available():
        try:
            f.factor(val)
            raised = False
        except api.SplltError as e:
            raised = e.flag == -30
        assert raised, "spllt_factor must fail loudly without a HIP device (no CPU fallback)"
    f.close()
    assert f.akeep.value is None and f.fkeep.value is None


def test_bad_arguments_set_flag_not_abort():
    lib = _lib.load()
    info = api.spllt_inform_t()
    lib.spllt_factor(None, None, None, 0, None, C.byref(info))
    assert info.flag == -10
    x = np.zeros(3)
    lib.spllt_solve(None, None, None, 1, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(info), 0)
    assert info.flag == -10
    tm = C.c_void_p(None)
    lib.spllt_task_manager_init(C.byref(tm))
    assert tm.value is not None
    st = C.c_int(7)
    lib.spllt_task_manager_deallocate(C.byref(tm), C.byref(st))
    assert tm.value is None and st.value == 0
    lib.spllt_wait()  # nothing pending: returns


def test_unknown_solve_job_is_rejected_like_reference():
    """src/spllt_solve_mod.F90:216-220: job outside 0..2 -> SPLLT_WARNING_PARAM_VALUE (-10);
    example/C/simple.c:69 passes 6 and gets exactly this."""
    A = matgen.poisson2d(6)
    n, ptr, row, val = api.csc_lower_1based(A)
    f = api.Factorization(n, ptr, row, nb=8)
    x = np.ones(n)
    f.lib.spllt_solve(f.fkeep, C.byref(f.options), f.order.ctypes.data_as(C.POINTER(C.c_int)), 1,
                      x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(f.info), 6)
    assert f.info.flag == -10


@pytest.mark.parametrize("name,ptr,row", [
    ("row-out-of-range", [1, 3, 5, 6], [1, 2, 2, 9, 3]),
    ("upper-triangle-entry", [1, 3, 5, 6], [1, 2, 1, 3, 3]),
    ("decreasing-ptr", [1, 3, 2, 6], [1, 2, 2, 3, 3]),
    ("duplicate-entry", [1, 4, 6, 7], [1, 2, 2, 2, 3, 3]),
    ("ptr-not-starting-at-1", [0, 2, 4, 5], [1, 2, 2, 3, 3]),
])
def test_analyse_rejects_malformed_patterns(name, ptr, row):
    """The reference passes ptr/row to SSIDS unchecked (undefined behaviour); the
    drop-in reports SPLLT_ERROR_PARAMETER (-10) instead of touching bad memory."""
    from spllt_amd import api
    with pytest.raises(api.SplltError) as ei:
        api.Factorization(3, np.array(ptr, dtype=np.int32), np.array(row, dtype=np.int32), nb=8)
    assert ei.value.flag == -10


def test_analyse_accepts_unsorted_rows_and_missing_diagonal():
    from spllt_amd import api
    f = api.Factorization(3, np.array([1, 4, 5, 6], dtype=np.int32),
                          np.array([3, 1, 2, 2, 3], dtype=np.int32), nb=8)
    assert f.sym_info()["nnz_l"] == 6
    g = api.Factorization(3, np.array([1, 2, 3, 4], dtype=np.int32), np.array([2, 3, 3], dtype=np.int32), nb=8)
    assert g.sym_info()["n"] == 3


def test_analyse_rejects_tile_sizes_beyond_the_kernel_limit():
    from spllt_amd import api
    ptr, row = np.array([1, 3, 5, 6], dtype=np.int32), np.array([1, 2, 2, 3, 3], dtype=np.int32)
    with pytest.raises(api.SplltError) as ei:
        api.Factorization(3, ptr, row, nb=2048)
    assert ei.value.flag == -98
    assert api.Factorization(3, ptr, row, nb=1024).sym_info()["n"] == 3


def test_wedged_runtime_makes_the_exit_handler_touch_nothing(tmp_path):
    """The wait / submission deadlines mark the HIP runtime as wedged, process-wide; from then on the
    library's atexit handler (events, pinned memory, cached device buffers: hipFree synchronises the
    whole device) must return at once, so that the process that detected a hang can still exit.
    Checked through the test hooks, in a process of its own: the handler's breadcrumbs say what it did."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    crumbs = tmp_path / "crumbs.txt"
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from spllt_amd import _lib\n"
        "lib = _lib.load()\n"
        "assert lib.spllt_hip_debug(b'wedged') == 0 and lib.spllt_hip_debug(b'nonsense') == -1\n"
        "lib.spllt_hip_debug(b'teardown')\n"
        "print('FIRST', open(%r).read().strip())\n"
        "assert lib.spllt_hip_debug(b'wedge') == 0 and lib.spllt_hip_debug(b'wedged') == 1\n"
        "lib.spllt_hip_debug(b'teardown')\n"
        "print('SECOND', open(%r).read().strip())\n"
    ) % (root, str(crumbs), str(crumbs))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, SPLLT_HIP_CRUMBS=str(crumbs)))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FIRST teardown: done" in r.stdout, r.stdout
    assert "SECOND teardown: skipped, the runtime is wedged" in r.stdout, r.stdout
