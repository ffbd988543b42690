"""Host-side mirror of SpLLT's user API over the C-ABI of libspllt_hip.so.

Function names, argument meaning and error behaviour follow the reference
(``spllt_analyse`` src/spllt_analyse_mod.F90:23, ``spllt_factor``
src/spllt_mod.F90:141, ``spllt_wait`` :172, ``spllt_solve``
src/spllt_solve_mod.F90:8-12; C forms include/spllt_iface.h:59-148):
1-based CSC of the lower triangle, status in ``info.flag``, factor is
asynchronous until ``wait``.  All numerical work happens behind the C-ABI in
hand-written HIP; this file only marshals arrays.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import spllt_hip_sym_info_t, spllt_inform_t, spllt_options_t

_I32 = ("order", "sptr", "sparent", "rlist", "small", "level", "bcol_node", "bcol_width",
        "bcol_r0", "bcol_nrow", "node_bcol0")
_I64 = ("rptr", "bcol_off", "map_dst", "map_src", "lmap_ptr", "weight")

UPD_UNIT_DTYPE = np.dtype([
    ("d_off", "<i8"), ("relrow_off", "<i8"), ("gcol_off", "<i8"), ("dinv_off", "<i8"),
    ("src_bcol0", "<i4"), ("nseg", "<i4"), ("seg_r0", "<i4"), ("seg_stride", "<i4"),
    ("src_r0", "<i4"), ("src_c0", "<i4"), ("M", "<i4"), ("N", "<i4"), ("k0", "<i4"),
    ("klen", "<i4"), ("d_ld", "<i4"), ("d_row0", "<i4"), ("d_col0", "<i4"), ("mode", "<i4"),
    ("dinv_ld", "<i4"), ("lower", "<i4"), ("b_bcol0", "<i4"), ("b_seg_r0", "<i4"),
    ("atomic", "<i4"), ("a_w", "<i4"), ("a_off", "<i8")])
UPD_TILE_DTYPE = np.dtype([("unit", "<i4"), ("ti", "<i2"), ("tj", "<i2")])
CHAIN_UNIT_DTYPE = np.dtype([("off", "<i8"), ("winv_off", "<i8"), ("ld", "<i4"), ("c0", "<i4"),
                             ("pn", "<i4"), ("cs", "<i4"), ("ce", "<i4"), ("gcol", "<i4")])
PANEL_UNIT_DTYPE = np.dtype([("off", "<i8"), ("dinv_off", "<i8"),
                             ("ld", "<i4"), ("c0", "<i4"), ("pn", "<i4"), ("next_pn", "<i4"),
                             ("nrow", "<i4"), ("gcol", "<i4"), ("ntile", "<i4"), ("pad_", "<i4")])
SUB_TASK_DTYPE = np.dtype([("g_off", "<i8"), ("node_first", "<i4"), ("node_count", "<i4"), ("g_n", "<i4"),
                           ("pad_", "<i4")])
SUB_NODE_DTYPE = np.dtype([("off", "<i8"), ("dinv_off", "<i8"), ("w", "<i4"), ("nrow", "<i4"), ("gcol", "<i4"),
                           ("unit_first", "<i4"), ("unit_count", "<i4"), ("root", "<i4")])
GATHER_ITEM_DTYPE = np.dtype([("buf_off", "<i8"), ("relrow_off", "<i8"), ("gcol_off", "<i8"), ("ld", "<i4"),
                              ("i0", "<i4"), ("i1", "<i4"), ("j0", "<i4"), ("j1", "<i4"),
                              ("diag_shift", "<i4"), ("lower", "<i4"), ("pad_", "<i4")])
GATHER_TILE_DTYPE = np.dtype([("d_off", "<i8"), ("d_ld", "<i4"), ("row0", "<i4"), ("col0", "<i4"),
                              ("rows", "<i4"), ("cols", "<i4"), ("drow_base", "<i4"),
                              ("dcol_base", "<i4"), ("first", "<i4"), ("count", "<i4"), ("pad_", "<i4")])
# "launches": int64 x 12 per launch
LAUNCH_COLS = ("kind", "level", "first", "count", "tile", "flops", "stream", "record",
               "wait0", "wait1", "wait2", "wait3")
SOLVE_UNIT_DTYPE = np.dtype([("off", "<i8"), ("dinv_off", "<i8"), ("idx_off", "<i8"), ("w", "<i4"),
                             ("nrow", "<i4"), ("pw", "<i4"), ("cb", "<i4"), ("gcol0", "<i4"), ("pad_", "<i4")])
POTRF_UNIT_DTYPE = np.dtype([("off", "<i8"), ("dinv_off", "<i8"), ("ld", "<i4"), ("n", "<i4"),
                             ("gcol", "<i4"), ("flags", "<i4")])


class SplltError(RuntimeError):
    def __init__(self, where, flag, msg=""):
        super().__init__(f"{where}: info.flag = {flag} {msg}".strip())
        self.flag = flag


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def csc_lower_1based(A):
    """scipy sparse (symmetric, any format) -> (n, ptr, row, val) of the lower
    triangle, 1-based int32, as the C-ABI expects (example/C/simple.c:37-44)."""
    import scipy.sparse as sp
    L = sp.tril(sp.csc_matrix(A), format="csc")
    L.sort_indices()
    return (L.shape[0], (L.indptr + 1).astype(np.int32), (L.indices + 1).astype(np.int32),
            np.ascontiguousarray(L.data, dtype=np.float64))


class Factorization:
    """One analysed pattern: owns the akeep/fkeep handle pair."""

    def __init__(self, n, ptr, row, nb=256, nemin=32, prune_tree=True, ncpu=1, order=None,
                 panel_width=None, tile=None, engine_flags=0, chain_block=None, symbolic=None):
        """symbolic: optional dict with the SSIDS-style quintuple (0-based numpy arrays
        "sptr", "sparent", "rptr", "rlist", "order"): the analyse then takes exactly this
        supernode partition and tree (spllt_hip_analyse_symbolic)."""
        self.lib = _lib.load()
        self.n = int(n)
        self.ptr = np.ascontiguousarray(ptr, dtype=np.int32)
        self.row = np.ascontiguousarray(row, dtype=np.int32)
        self.nnz = int(self.ptr[n] - 1) if n > 0 else 0
        self.options = spllt_options_t.default()
        self.options.nb = nb
        self.options.nemin = nemin
        self.options.prune_tree = 1 if prune_tree else 0
        self.options.ncpu = ncpu
        self.akeep = C.c_void_p(None)
        self.fkeep = C.c_void_p(None)
        self.info = spllt_inform_t()
        self.order = np.zeros(max(n, 1), dtype=np.int32)
        if symbolic is not None:
            sy = symbolic
            sp = np.ascontiguousarray(np.asarray(sy["sptr"]) + 1, dtype=np.int32)
            spar = np.ascontiguousarray(np.asarray(sy["sparent"]) + 1, dtype=np.int32)
            rp = np.ascontiguousarray(np.asarray(sy["rptr"]) + 1, dtype=np.int64)
            rl = np.ascontiguousarray(np.asarray(sy["rlist"]) + 1, dtype=np.int32)
            oin = np.ascontiguousarray(np.asarray(sy["order"]) + 1, dtype=np.int32)
            self.lib.spllt_hip_analyse_symbolic(C.byref(self.akeep), C.byref(self.fkeep),
                                                C.byref(self.options), n, _ip(self.ptr), _ip(self.row),
                                                C.byref(self.info), len(spar), _ip(sp), _ip(spar),
                                                rp.ctypes.data_as(C.POINTER(C.c_int64)), _ip(rl), _ip(oin))
            self.order[:n] = oin[:n]
        elif order is None:
            self.lib.spllt_analyse(C.byref(self.akeep), C.byref(self.fkeep), C.byref(self.options),
                                   n, _ip(self.ptr), _ip(self.row), C.byref(self.info),
                                   _ip(self.order))
        else:
            oin = np.ascontiguousarray(order, dtype=np.int32)
            self.lib.spllt_hip_analyse_ordered(C.byref(self.akeep), C.byref(self.fkeep),
                                               C.byref(self.options), n, _ip(self.ptr),
                                               _ip(self.row), C.byref(self.info), _ip(self.order),
                                               _ip(oin))
        if self.info.flag < 0:
            raise SplltError("spllt_analyse", self.info.flag)
        if panel_width or tile or engine_flags:
            self.lib.spllt_hip_set_engine(self.fkeep, panel_width or 0, tile or 0, engine_flags)
        if chain_block:
            self.lib.spllt_hip_set_chain_block(self.fkeep, int(chain_block))
        self._val_keepalive = None

    # ---- symbolic introspection ------------------------------------------
    def sym_info(self):
        si = spllt_hip_sym_info_t()
        rc = self.lib.spllt_hip_sym_info(self.akeep, C.byref(si))
        if rc:
            raise SplltError("spllt_hip_sym_info", rc)
        d = {k: getattr(si, k) for k, _ in si._fields_}
        d["ordering"] = si.ordering.decode()
        return d

    def sym(self, name):
        dt = np.int32 if name in _I32 else np.int64
        if name not in _I32 and name not in _I64:
            raise KeyError(name)
        cnt = self.lib.spllt_hip_sym_get(self.akeep, name.encode(), None, 0)
        if cnt < 0:
            raise KeyError(name)
        out = np.zeros(max(cnt, 1), dtype=dt)
        self.lib.spllt_hip_sym_get(self.akeep, name.encode(), out.ctypes.data, cnt)
        return out[:cnt]

    def program(self, name):
        nbytes = self.lib.spllt_hip_program_get(self.fkeep, name.encode(), None, 0)
        if nbytes < 0:
            raise KeyError(name)
        raw = np.zeros(max(nbytes, 1), dtype=np.uint8)
        self.lib.spllt_hip_program_get(self.fkeep, name.encode(), raw.ctypes.data, nbytes)
        raw = raw[:nbytes]
        if name == "launches":
            return raw.view(np.int64).reshape(-1, len(LAUNCH_COLS))
        if name == "units":
            return raw.view(UPD_UNIT_DTYPE)
        if name == "tiles":
            return raw.view(UPD_TILE_DTYPE)
        if name == "potrf":
            return raw.view(POTRF_UNIT_DTYPE)
        if name == "chains":
            return raw.view(CHAIN_UNIT_DTYPE)
        if name == "panels":
            return raw.view(PANEL_UNIT_DTYPE)
        if name == "sub_tasks":
            return raw.view(SUB_TASK_DTYPE)
        if name == "sub_nodes":
            return raw.view(SUB_NODE_DTYPE)
        if name == "exchanges":
            return raw.view(np.int64).reshape(-1, 5)
        if name == "xitems":
            return raw.view(np.int64).reshape(-1, 6)
        if name == "xbuf_elems":
            return int(raw.view(np.int64)[0])
        if name in ("chain_block", "scratch_size", "panel_width", "gen_size"):
            return int(raw.view(np.int64)[0])
        if name == "gather_tiles":
            return raw.view(GATHER_TILE_DTYPE)
        if name == "gather_items":
            return raw.view(GATHER_ITEM_DTYPE)
        if name == "relpos":
            return raw.view(np.int32)
        if name == "dinv_size":
            return int(raw.view(np.int64)[0])
        if name == "solve_units":
            return raw.view(SOLVE_UNIT_DTYPE)
        if name == "solve_list":
            return raw.view(np.int32)
        if name == "solve_tiles":
            return raw.view(UPD_TILE_DTYPE)
        if name in ("solve_fwd", "solve_bwd"):
            return raw.view(np.int64).reshape(-1, 4)
        if name == "solve_split":
            return raw.view(np.int64)
        return raw

    # ---- numerical phases --------------------------------------------------
    def factor(self, val):
        """spllt_factor: asynchronous; call wait() before using L."""
        val = np.ascontiguousarray(val, dtype=np.float64)
        self._val_keepalive = val  # `val` must outlive the submission
        self.lib.spllt_factor(self.akeep, self.fkeep, C.byref(self.options), self.nnz, _dp(val),
                              C.byref(self.info))
        if self.info.flag < 0:
            raise SplltError("spllt_factor", self.info.flag, self.last_error())
        return self

    def factor_dev(self, val_dev_ptr):
        """spllt_factor with `val` already in HBM (integer device pointer)."""
        self.lib.spllt_hip_factor_dev(self.akeep, self.fkeep, C.byref(self.options), self.nnz,
                                      C.c_void_p(val_dev_ptr), C.byref(self.info))
        if self.info.flag < 0:
            raise SplltError("spllt_factor", self.info.flag, self.last_error())
        return self

    def wait(self):
        rc = self.lib.spllt_hip_wait(self.fkeep)
        self._val_keepalive = None
        if rc < 0:
            raise SplltError("spllt_wait", rc, self.last_error())
        return self

    def times(self):
        s, d, h = C.c_double(), C.c_double(), C.c_double()
        nl = C.c_int()
        self.lib.spllt_hip_factor_times(self.fkeep, C.byref(s), C.byref(d), C.byref(h), C.byref(nl))
        return {"submit_ms": s.value, "device_ms": d.value, "h2d_ms": h.value,
                "launches": nl.value}

    def get_factor(self, out=None):
        """the factor's arena on the host; out: an existing float64 array of that length to fill
        (a fresh one pays its page faults during the copy)"""
        arena = self.sym_info()["arena"]
        if out is None:
            out = np.zeros(max(arena, 1), dtype=np.float64)
        assert out.dtype == np.float64 and out.size >= arena and out.flags["C_CONTIGUOUS"]
        rc = self.lib.spllt_hip_get_factor(self.fkeep, _dp(out), arena)
        if rc < 0:
            raise SplltError("spllt_hip_get_factor", rc, self.last_error())
        return out[:arena]

    def device_factor_ptr(self):
        return self.lib.spllt_hip_device_factor(self.fkeep)

    def solve(self, b, job=0):
        """spllt_solve on a copy of b (n or n x nrhs, column-major per rhs)."""
        x = np.array(b, dtype=np.float64, order="F", copy=True)
        nrhs = 1 if x.ndim == 1 else x.shape[1]
        ws = C.c_long()
        self.lib.spllt_prepare_solve(self.akeep, self.fkeep, self.options.nb, nrhs, C.byref(ws),
                                     C.byref(self.info))
        self.lib.spllt_solve(self.fkeep, C.byref(self.options), _ip(self.order), nrhs, _dp(x),
                             C.byref(self.info), job)
        if self.info.flag < 0:
            raise SplltError("spllt_solve", self.info.flag, self.last_error())
        return x

    def solve_dev(self, y_dev_ptr, nrhs=1, job=0, phase=-1):
        """spllt_hip_solve_dev: substitution on device vectors in pivot order
        (y[q*n + p(i)] = b_q[i], p = 0-based pivot position ("order" of spllt_hip_sym_get)), in place; phase 0/1/2 on a partitioned factor."""
        rc = self.lib.spllt_hip_solve_dev(self.fkeep, C.c_void_p(y_dev_ptr), nrhs, job, phase)
        if rc < 0:
            raise SplltError("spllt_hip_solve_dev", rc, self.last_error())
        return self

    # ---- multi-GPU subtree partition ------------------------------------------
    def set_partition(self, rank, nranks):
        """Declare this process as `rank` of `nranks`; returns the number of
        doubles of the top-tree exchange buffer (0 for nranks == 1)."""
        n = C.c_int64()
        rc = self.lib.spllt_hip_set_partition(self.fkeep, rank, nranks, C.byref(n))
        if rc < 0:
            raise SplltError("spllt_hip_set_partition", rc)
        self.rank, self.nranks = rank, nranks
        return n.value

    def set_communicator(self, nccl_comm):
        """hand the caller's RCCL communicator (ncclComm_t as an integer) to the library: the
        exchanges of the partition then run inside spllt_factor / spllt_wait / spllt_solve"""
        rc = self.lib.spllt_hip_set_communicator(self.fkeep, C.c_void_p(nccl_comm))
        if rc < 0:
            raise SplltError("spllt_hip_set_communicator", rc, self.last_error())

    def set_exchange_buffer(self, dev_ptr):
        rc = self.lib.spllt_hip_set_exchange_buffer(self.fkeep, C.c_void_p(dev_ptr))
        if rc < 0:
            raise SplltError("spllt_hip_set_exchange_buffer", rc)

    def engine_stream(self):
        """hipStream_t (integer) on which the exchange buffer is packed / unpacked"""
        p = self.lib.spllt_hip_engine_stream(self.fkeep)
        if not p:
            raise SplltError("spllt_hip_engine_stream", -30, self.last_error())
        return int(p)

    def exchange_stream(self):
        """hipStream_t (integer) of the pending exchange: its collective belongs on this stream"""
        return int(self.lib.spllt_hip_exchange_stream(self.fkeep) or 0)

    def continue_after_exchange(self):
        rc = self.lib.spllt_hip_continue(self.fkeep)
        if rc < 0:
            raise SplltError("spllt_hip_continue", rc, self.last_error())
        return self

    def pending_exchange(self):
        """index of the exchange (program("exchanges")) the engine is waiting for, -1: none"""
        return int(self.lib.spllt_hip_pending_exchange(self.fkeep))

    def partition(self, name):
        nbytes = self.lib.spllt_hip_partition_get(self.fkeep, name.encode(), None, 0)
        if nbytes < 0:
            raise KeyError(name)
        raw = np.zeros(max(nbytes, 1), dtype=np.uint8)
        self.lib.spllt_hip_partition_get(self.fkeep, name.encode(), raw.ctypes.data, nbytes)
        raw = raw[:nbytes]
        if name == "arena_elems":
            return raw.view(np.int64)
        return raw if name == "map_keep" else raw.view(np.int32)

    def profile(self, val, in_program=False):
        """per-launch device time (ms) of one factorization: launches serialized on one
        stream (kernels alone on the chip), or in_program=True: inside the real
        multi-stream program (durations under contention)"""
        val = np.ascontiguousarray(val, dtype=np.float64)
        nl = len(self.program("launches"))
        ms = np.zeros(max(nl, 1), dtype=np.float32)
        fn = self.lib.spllt_hip_profile_in_program if in_program else self.lib.spllt_hip_profile
        rc = fn(self.fkeep, _dp(val), self.nnz, ms.ctypes.data_as(C.POINTER(C.c_float)), nl)
        if rc < 0:
            raise SplltError("spllt_hip_profile", rc, self.last_error())
        return ms[:rc]

    def timeline(self, val):
        """spllt_hip_timeline: ms after the value scatter at which the event of every recording
        launch of the real multi-stream program completed (-1: none); last entry = the end"""
        val = np.ascontiguousarray(val, dtype=np.float64)
        nl = len(self.program("launches")) + 1
        t = np.zeros(nl, dtype=np.float32)
        rc = self.lib.spllt_hip_timeline(self.fkeep, _dp(val), self.nnz, t.ctypes.data_as(C.POINTER(C.c_float)), nl)
        if rc < 0:
            raise SplltError("spllt_hip_timeline", rc, self.last_error())
        return t[:rc]

    def last_error(self):
        return (self.lib.spllt_hip_last_error(self.fkeep) or b"").decode()

    def close(self):
        if getattr(self, "lib", None) is None:
            return
        st = C.c_int()
        if self.fkeep:
            self.lib.spllt_deallocate_fkeep(C.byref(self.fkeep), C.byref(st))
        if self.akeep:
            self.lib.spllt_deallocate_akeep(C.byref(self.akeep), C.byref(st))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def residual(n, ptr, row, val, x, b):
    """||A x - b||_2 / ||b||_2 for the symmetric matrix given by its 1-based
    CSC lower triangle (the metric of reference drivers/spllt_omp.F90:248-262)."""
    import scipy.sparse as sp
    L = sp.csc_matrix((val, np.asarray(row) - 1, np.asarray(ptr) - 1), shape=(n, n))
    A = L + sp.tril(L, -1).T
    r = A @ x - b
    return float(np.linalg.norm(r) / np.linalg.norm(b))
