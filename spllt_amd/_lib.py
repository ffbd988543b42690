"""ctypes binding of libspllt_hip.so (the C-ABI of include/spllt_iface.h + spllt_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C spllt_amd/csrc``.  There is no Python or CPU fallback for the
factorize path: if the shared library is missing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspllt_hip.so")


class spllt_options_t(C.Structure):
    """reference include/spllt_iface.h:14-31"""
    _fields_ = [(k, C.c_int) for k in (
        "print_level", "nrhs", "ncpu", "nb", "nemin", "prune_tree", "min_width_blas",
        "nb_min", "nb_max", "nrhs_min", "nrhs_max", "nb_linear_comp", "nrhs_linear_comp",
        "chunk")]

    @classmethod
    def default(cls):
        """SPLLT_OPTIONS_NULL(), reference include/spllt_iface.h:33-47"""
        return cls(print_level=0, nrhs=1, ncpu=1, nb=16, nemin=32, prune_tree=1,
                   min_width_blas=8, nb_min=32, nb_max=32, nrhs_min=1, nrhs_max=1,
                   nb_linear_comp=0, nrhs_linear_comp=0, chunk=10)


class spllt_inform_t(C.Structure):
    """reference include/spllt_iface.h:49-57"""
    _fields_ = [(k, C.c_int) for k in (
        "flag", "maxdepth", "num_factor", "num_flops", "num_nodes", "stat")]


class spllt_hip_sym_info_t(C.Structure):
    _fields_ = [(k, C.c_int64) for k in (
        "n", "nnz_a", "nnodes", "nbcol", "nblk", "arena", "nnz_l", "flops", "rlist_len")] + \
        [(k, C.c_int) for k in ("nb", "maxmn", "maxdepth", "nlevels")] + \
        [("ordering", C.c_char * 16)]


# every symbol declared in include/spllt_iface.h and include/spllt_hip.h
IFACE_SYMBOLS = [
    "spllt_analyse", "spllt_factor", "spllt_prepare_solve", "spllt_set_mem_solve",
    "spllt_solve_workspace_size", "spllt_solve", "spllt_solve_worker", "spllt_wait",
    "spllt_chkerr", "spllt_deallocate_fkeep", "spllt_deallocate_akeep",
    "spllt_task_manager_deallocate", "spllt_task_manager_init", "spllt_all",
]
HIP_SYMBOLS = [
    "spllt_factor_diag_block_hip", "spllt_solve_block_hip", "spllt_update_block_hip",
    "spllt_update_between_hip", "spllt_expand_buffer_hip", "spllt_scatter_block_hip",
    "spllt_init_lfact_hip", "spllt_hip_analyse_ordered", "spllt_hip_sym_info",
    "spllt_hip_sym_get", "spllt_hip_set_engine", "spllt_hip_factor_dev", "spllt_hip_wait",
    "spllt_hip_get_factor", "spllt_hip_device_factor", "spllt_hip_factor_times",
    "spllt_hip_program_get", "spllt_hip_profile", "spllt_hip_last_error", "spllt_hip_version",
    "spllt_hip_set_partition", "spllt_hip_set_exchange_buffer", "spllt_hip_continue",
    "spllt_hip_pending_exchange",
    "spllt_hip_partition_get", "spllt_hip_solve_dev", "spllt_hip_set_chain_block", "spllt_hip_engine_stream", "spllt_hip_analyse_symbolic", "spllt_hip_profile_in_program", "spllt_hip_timeline",
    "spllt_hip_read_rb", "spllt_hip_read_mm", "spllt_hip_free_matrix", "spllt_hip_set_communicator",
    "spllt_hip_last_flag", "spllt_hip_debug", "spllt_hip_exchange_stream",
]

_lib = None


def load():
    """Load libspllt_hip.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C spllt_amd/csrc`. spllt_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    opt, inf = C.POINTER(spllt_options_t), C.POINTER(spllt_inform_t)
    lib.spllt_analyse.argtypes = [vpp, vpp, opt, C.c_int, ip, ip, inf, ip]
    lib.spllt_hip_debug.argtypes = [C.c_char_p]
    lib.spllt_hip_debug.restype = C.c_int
    lib.spllt_analyse.restype = None
    lib.spllt_hip_analyse_ordered.argtypes = [vpp, vpp, opt, C.c_int, ip, ip, inf, ip, ip]
    lib.spllt_hip_analyse_ordered.restype = None
    lib.spllt_hip_analyse_symbolic.argtypes = [vpp, vpp, opt, C.c_int, ip, ip, inf, C.c_int, ip, ip,
                                               C.POINTER(C.c_int64), ip, ip]
    lib.spllt_hip_analyse_symbolic.restype = None
    lib.spllt_factor.argtypes = [vp, vp, opt, C.c_int, dp, inf]
    lib.spllt_factor.restype = None
    lib.spllt_hip_factor_dev.argtypes = [vp, vp, opt, C.c_int, vp, inf]
    lib.spllt_hip_factor_dev.restype = None
    lib.spllt_prepare_solve.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_long), inf]
    lib.spllt_prepare_solve.restype = None
    lib.spllt_set_mem_solve.argtypes = [vp, vp, C.c_int, C.c_int, C.c_long, dp, dp, inf]
    lib.spllt_set_mem_solve.restype = None
    lib.spllt_solve_workspace_size.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_long)]
    lib.spllt_solve_workspace_size.restype = None
    lib.spllt_solve.argtypes = [vp, opt, ip, C.c_int, dp, inf, C.c_int]
    lib.spllt_solve.restype = None
    lib.spllt_solve_worker.argtypes = [vp, opt, ip, C.c_int, dp, inf, C.c_int, dp, C.c_long, vp]
    lib.spllt_solve_worker.restype = None
    lib.spllt_wait.argtypes = []
    lib.spllt_wait.restype = None
    lib.spllt_chkerr.argtypes = [C.c_int, ip, ip, dp, C.c_int, dp, dp]
    lib.spllt_chkerr.restype = None
    lib.spllt_deallocate_fkeep.argtypes = [vpp, ip]
    lib.spllt_deallocate_fkeep.restype = None
    lib.spllt_deallocate_akeep.argtypes = [vpp, ip]
    lib.spllt_deallocate_akeep.restype = None
    lib.spllt_task_manager_init.argtypes = [vpp]
    lib.spllt_task_manager_init.restype = None
    lib.spllt_task_manager_deallocate.argtypes = [vpp, ip]
    lib.spllt_task_manager_deallocate.restype = None
    lib.spllt_all.argtypes = [vpp, vpp, opt, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, dp, dp,
                              dp, inf]
    lib.spllt_all.restype = None
    lib.spllt_hip_sym_info.argtypes = [vp, C.POINTER(spllt_hip_sym_info_t)]
    lib.spllt_hip_sym_info.restype = C.c_int
    lib.spllt_hip_sym_get.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.spllt_hip_sym_get.restype = C.c_int64
    lib.spllt_hip_set_engine.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.spllt_hip_set_engine.restype = C.c_int
    lib.spllt_hip_engine_stream.argtypes = [vp]
    lib.spllt_hip_engine_stream.restype = vp
    lib.spllt_hip_exchange_stream.argtypes = [vp]
    lib.spllt_hip_exchange_stream.restype = vp
    lib.spllt_hip_set_chain_block.argtypes = [vp, C.c_int]
    lib.spllt_hip_set_chain_block.restype = C.c_int
    lib.spllt_hip_wait.argtypes = [vp]
    lib.spllt_hip_wait.restype = C.c_int
    lib.spllt_hip_get_factor.argtypes = [vp, dp, C.c_int64]
    lib.spllt_hip_get_factor.restype = C.c_int
    lib.spllt_hip_device_factor.argtypes = [vp]
    lib.spllt_hip_device_factor.restype = C.c_void_p
    lib.spllt_hip_factor_times.argtypes = [vp, dp, dp, dp, ip]
    lib.spllt_hip_factor_times.restype = C.c_int
    lib.spllt_hip_program_get.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.spllt_hip_program_get.restype = C.c_int64
    lib.spllt_hip_set_partition.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    lib.spllt_hip_set_partition.restype = C.c_int
    lib.spllt_hip_set_exchange_buffer.argtypes = [vp, vp]
    lib.spllt_hip_set_exchange_buffer.restype = C.c_int
    lib.spllt_hip_pending_exchange.argtypes = [vp]
    lib.spllt_hip_pending_exchange.restype = C.c_int
    lib.spllt_hip_continue.argtypes = [vp]
    lib.spllt_hip_continue.restype = C.c_int
    lib.spllt_hip_partition_get.argtypes = [vp, C.c_char_p, vp, C.c_int64]
    lib.spllt_hip_partition_get.restype = C.c_int64
    lib.spllt_hip_solve_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.spllt_hip_solve_dev.restype = C.c_int
    lib.spllt_hip_profile.argtypes = [vp, dp, C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.spllt_hip_profile.restype = C.c_int
    lib.spllt_hip_profile_in_program.argtypes = [vp, dp, C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.spllt_hip_profile_in_program.restype = C.c_int
    lib.spllt_hip_timeline.argtypes = [vp, dp, C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.spllt_hip_timeline.restype = C.c_int
    lib.spllt_hip_last_error.argtypes = [vp]
    lib.spllt_hip_last_error.restype = C.c_char_p
    lib.spllt_hip_version.argtypes = []
    lib.spllt_hip_version.restype = C.c_char_p
    # kernel operators (device pointers passed as integers)
    lib.spllt_factor_diag_block_hip.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.spllt_factor_diag_block_hip.restype = C.c_int
    lib.spllt_solve_block_hip.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.spllt_solve_block_hip.restype = C.c_int
    lib.spllt_update_block_hip.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp]
    lib.spllt_update_block_hip.restype = C.c_int
    lib.spllt_update_between_hip.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int,
                                             vp, vp, C.c_int]
    lib.spllt_update_between_hip.restype = C.c_int
    lib.spllt_expand_buffer_hip.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp]
    lib.spllt_expand_buffer_hip.restype = C.c_int
    lib.spllt_scatter_block_hip.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp,
                                            C.c_int, vp, C.c_int, vp, C.c_int]
    lib.spllt_scatter_block_hip.restype = C.c_int
    lib.spllt_init_lfact_hip.argtypes = [vp, vp, vp, vp, vp, C.c_int64]
    lib.spllt_init_lfact_hip.restype = C.c_int
    ip, ipp, dpp = C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_double))
    for fn in (lib.spllt_hip_read_rb, lib.spllt_hip_read_mm):
        fn.argtypes = [C.c_char_p, C.c_int, C.c_int, ip, ip, ipp, ipp, dpp]
        fn.restype = C.c_int
    lib.spllt_hip_free_matrix.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.spllt_hip_free_matrix.restype = None
    lib.spllt_hip_set_communicator.argtypes = [vp, vp]
    lib.spllt_hip_set_communicator.restype = C.c_int
    lib.spllt_hip_last_flag.argtypes = [vp]
    lib.spllt_hip_last_flag.restype = C.c_int
    _lib = lib
    return lib
