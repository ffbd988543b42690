"""ctypes wrapper of the CPU oracle (oracle/libspllt_oracle*.so).

TEST INFRASTRUCTURE ONLY (see oracle/spllt_oracle.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by spllt_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


class spo_block(C.Structure):
    _fields_ = [("id", C.c_int64), ("dblk", C.c_int64), ("last_blk", C.c_int64), ("sa", C.c_int64),
                ("bcol", C.c_int), ("blkm", C.c_int), ("blkn", C.c_int), ("node", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


def load(variant="plain"):
    """variant: 'plain' (C loops, sequential) or 'mkl' (vendor BLAS + OpenMP tasks)."""
    if variant in _libs:
        return _libs[variant]
    name = "libspllt_oracle.so" if variant == "plain" else "libspllt_oracle_mkl.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not available")
    if variant != "plain":
        # libmkl_rt picks its layers at first call: one BLAS thread per caller,
        # parallelism comes from the oracle's OpenMP tasks like in the reference.
        os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
        os.environ.setdefault("MKL_INTERFACE_LAYER", "LP64")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    i32p, i64p, dp, vp = (C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                          C.c_void_p)
    lib.spo_create.argtypes = [C.c_int, C.c_int, i32p, i32p, i64p, i32p, i32p, i64p, i32p, C.c_int,
                               i32p, C.c_int]
    lib.spo_create.restype = vp
    lib.spo_destroy.argtypes = [vp]
    lib.spo_factorize.argtypes = [vp, dp, C.c_int]
    lib.spo_factorize.restype = C.c_int
    lib.spo_solve.argtypes = [vp, C.c_int, dp]
    lib.spo_nbcol.argtypes = [vp]
    lib.spo_nbcol.restype = C.c_int
    lib.spo_nblk.argtypes = [vp]
    lib.spo_nblk.restype = C.c_int64
    lib.spo_maxmn.argtypes = [vp]
    lib.spo_maxmn.restype = C.c_int
    lib.spo_arena.argtypes = [vp]
    lib.spo_arena.restype = C.c_int64
    lib.spo_lcol_size.argtypes = [vp, C.c_int]
    lib.spo_lcol_size.restype = C.c_int64
    lib.spo_blocks.argtypes = [vp]
    lib.spo_blocks.restype = C.POINTER(spo_block)
    lib.spo_lmap_len.argtypes = [vp, C.c_int]
    lib.spo_lmap_len.restype = C.c_int64
    lib.spo_lmap_dst.argtypes = [vp, C.c_int]
    lib.spo_lmap_dst.restype = i64p
    lib.spo_lmap_src.argtypes = [vp, C.c_int]
    lib.spo_lmap_src.restype = i64p
    lib.spo_export_arena.argtypes = [vp, dp]
    lib.spo_blas_name.restype = C.c_char_p
    lib.spo_factor_diag_block.argtypes = [C.c_int, C.c_int, dp]
    lib.spo_factor_diag_block.restype = C.c_int
    lib.spo_solve_block.argtypes = [C.c_int, C.c_int, dp, dp]
    lib.spo_update_block.argtypes = [C.c_int, C.c_int, dp, C.c_int, C.c_int, dp, dp]
    lib.spo_expand_buffer.argtypes = [dp, C.c_int, i32p, C.c_int, i32p, C.c_int, C.c_int, dp]
    lib.spo_update_direct.argtypes = [C.c_int, dp, C.c_int, dp, dp, i32p, C.c_int, i32p, C.c_int,
                                      C.c_int]
    lib.spo_scatter_block.argtypes = [C.c_int, C.c_int, i32p, i32p, dp, C.c_int, i32p, i32p, dp,
                                      C.c_int]
    lib.spo_compute_map.argtypes = [C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p, i32p,
                                    i32p, i32p, i32p, i32p, i32p, i32p]
    lib.spo_compute_map.restype = C.c_int
    _libs[variant] = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OracleFactor:
    """The oracle's view of one analysed pattern.  Inputs are 0-based numpy
    arrays: the SSIDS-style quintuple + the user's CSC-lower pattern."""

    def __init__(self, n, sptr, sparent, rptr, rlist, order, ptr0, row0, nb, small=None,
                 min_width_blas=8, variant="plain"):
        self.lib = load(variant)
        self.n = n
        self.keep = [np.ascontiguousarray(sptr, np.int32), np.ascontiguousarray(sparent, np.int32),
                     np.ascontiguousarray(rptr, np.int64), np.ascontiguousarray(rlist, np.int32),
                     np.ascontiguousarray(order, np.int32), np.ascontiguousarray(ptr0, np.int64),
                     np.ascontiguousarray(row0, np.int32)]
        sm = None if small is None else np.ascontiguousarray(small, np.int32)
        self.keep.append(sm)
        k = self.keep
        self.h = self.lib.spo_create(
            n, len(k[0]) - 1, _p(k[0], C.c_int), _p(k[1], C.c_int), _p(k[2], C.c_int64),
            _p(k[3], C.c_int), _p(k[4], C.c_int), _p(k[5], C.c_int64), _p(k[6], C.c_int), nb,
            None if sm is None else _p(sm, C.c_int), min_width_blas)

    @classmethod
    def from_factorization(cls, f, small=None, min_width_blas=8, variant="plain"):
        """Build from a spllt_amd.api.Factorization's symbolic output (so that
        both sides factorize the identical (nodes, bc, lmap))."""
        return cls(f.n, f.sym("sptr"), f.sym("sparent"), f.sym("rptr"), f.sym("rlist"),
                   f.sym("order"), f.ptr.astype(np.int64) - 1, f.row - 1, f.sym_info()["nb"],
                   small=small, min_width_blas=min_width_blas, variant=variant)

    def factorize(self, val, nthreads=1):
        val = np.ascontiguousarray(val, np.float64)
        return self.lib.spo_factorize(self.h, _p(val, C.c_double), nthreads)

    def arena(self):
        out = np.zeros(max(self.lib.spo_arena(self.h), 1))
        self.lib.spo_export_arena(self.h, _p(out, C.c_double))
        return out[:self.lib.spo_arena(self.h)]

    def solve(self, b):
        x = np.array(b, dtype=np.float64, order="F", copy=True)
        nrhs = 1 if x.ndim == 1 else x.shape[1]
        self.lib.spo_solve(self.h, nrhs, _p(x, C.c_double))
        return x

    def blocks(self):
        nb = self.lib.spo_nblk(self.h)
        arr = self.lib.spo_blocks(self.h)
        return [(arr[i].id, arr[i].dblk, arr[i].last_blk, arr[i].sa, arr[i].bcol, arr[i].blkm,
                 arr[i].blkn, arr[i].node) for i in range(nb)]

    def lmap(self, b):
        k = self.lib.spo_lmap_len(self.h, b)
        d = np.ctypeslib.as_array(self.lib.spo_lmap_dst(self.h, b), shape=(max(k, 1),))[:k].copy()
        s = np.ctypeslib.as_array(self.lib.spo_lmap_src(self.h, b), shape=(max(k, 1),))[:k].copy()
        return d, s

    def close(self):
        if self.h:
            self.lib.spo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
