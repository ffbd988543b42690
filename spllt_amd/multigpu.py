"""Multi-GPU factorization: one process per GPU, subtree partition, RCCL exchanges on the
engine's stream.

Every rank analyses the same pattern and maps the assembly tree onto the ranks
proportionally (symbolic.cpp, assign_owners: whole branches per rank, the nodes
whose subtrees span several ranks form the top tree - the role the
pruning layer of spllt_prune_tree, reference src/spllt_analyse_mod.F90:806-987,
plays for the reference's sequential subtrees), owns the branches assigned to it,
and accumulates its contributions to the
top tree in its own (zero-initialised) copy of the top-tree block columns.  The
extend-add of the reference -- generated element + spllt_scatter_block
(src/spllt_factorization_mod.F90:39-191, src/spllt_kernels_mod.F90:1122-1160)
-- becomes a collective over xGMI, enqueued on the engine's own stream (no host
synchronisation between the phases):
  * replicated top tree: ONE all-reduce(sum); the top tree is then factorized on every rank;
  * distributed top tree (the reference's PaRSEC build distributes blocks 1-D cyclically,
    src/PaRSEC/spllt_parsec_blk_data.c:33-64): ONE reduce-scatter(sum) to the owners of the
    top-tree block columns, then one broadcast per finished block-column step; every rank
    updates only the destination block columns it owns.
The engine's program lists the exchanges (exchange_plan); run_exchange is the collective of one.

torch is used only for device memory and torch.distributed (plumbing).
"""
import os
import sys
import time

import numpy as np


def reduce_exchange_buffer(xbuf, group=None):
    """Sum the packed top-tree block columns over the ranks of `group` (in place).
    backend nccl == RCCL on ROCm; the same call runs on gloo for the CPU tests."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(xbuf, op=dist.ReduceOp.SUM, group=group)
    return xbuf


X_REDUCE_ALL, X_REDUCE_OWNER, X_BCAST, X_FLAG = 0, 1, 2, 3


def exchange_plan(f):
    """Per exchange of f's program (include/spllt_hip.h, spllt_hip_set_partition):
    (kind, elems, chunk, segments) with segments = [(root, offset, length)] of an X_BCAST.
    An X_REDUCE_OWNER (one per level of a distributed top tree) covers the REGION
    [elems - world * chunk, elems) of the buffer."""
    plan = []
    items = f.program("xitems")
    for kind, first, n, elems, chunk in f.program("exchanges").tolist():
        segs = []
        if kind == X_BCAST:
            for b, root, xo, cnt, _off, _space in items[first:first + n].tolist():
                if segs and segs[-1][0] == root and segs[-1][1] + segs[-1][2] == xo:
                    segs[-1][2] += cnt
                else:
                    segs.append([root, xo, cnt])
        plan.append((kind, elems, chunk, [tuple(s) for s in segs]))
    return plan


def run_exchange(xbuf, step, rank, world, group=None, scratch=None, force=False):
    """The collective of one exchange on the (torch) exchange buffer, in place; enqueue-only on
    RCCL.  step = one entry of exchange_plan().  X_REDUCE_OWNER leaves rank r's sum at
    xbuf[r*chunk:(r+1)*chunk] (gloo has no reduce-scatter: an all-reduce does the same there)."""
    import torch.distributed as dist
    kind, elems, chunk, segs = step
    if not dist.is_initialized() or (dist.get_world_size(group) <= 1 and not force):
        return xbuf       # (force: tests run the calls on a one-rank RCCL group)
    if kind in (X_REDUCE_ALL, X_FLAG):
        dist.all_reduce(xbuf[:elems], op=dist.ReduceOp.SUM, group=group)
    elif kind == X_REDUCE_OWNER:
        base = elems - world * chunk           # the region of this level's exchange
        if dist.get_backend(group) == "nccl":
            out = scratch[:chunk] if scratch is not None else xbuf.new_empty(chunk)
            dist.reduce_scatter_tensor(out, xbuf[base:elems], op=dist.ReduceOp.SUM, group=group)
            xbuf[base + rank * chunk:base + (rank + 1) * chunk].copy_(out)
        else:
            dist.all_reduce(xbuf[base:elems], op=dist.ReduceOp.SUM, group=group)
    elif kind == X_BCAST:
        for root, off, cnt in segs:
            src = dist.get_global_rank(group, root) if group is not None else root
            dist.broadcast(xbuf[off:off + cnt], src=src, group=group)
    return xbuf


class _NcclUniqueId(__import__("ctypes").Structure):
    _fields_ = [("internal", __import__("ctypes").c_char * 128)]


def make_rccl_communicator(rank, world, group=None):
    """An ncclComm_t of this process's own librccl (the copy torch loaded), for
    spllt_hip_set_communicator -- what a C or Fortran caller of the library would hand over (the
    reference keeps its distributed hook inside the library too, src/PaRSEC/spllt_parsec_blk_data.c:33-64).
    Rank 0 draws the unique id, the process group (any backend) ships its 128 bytes, every rank
    calls ncclCommInitRank.  Returns (comm handle as int, destroy function)."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    uid = _NcclUniqueId()
    if rank == 0:
        rc = rccl.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId failed: {rc}")
    if world > 1:
        box = [bytes(uid.internal) if rank == 0 else None]       # (c_char array: raw bytes, zeros included)
        if rank == 0:
            box = [C.string_at(C.addressof(uid), 128)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.addressof(uid), box[0], 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError(f"ncclCommInitRank({world}, rank {rank}) failed: {rc}")
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]

    def destroy():
        rccl.ncclCommDestroy(comm)
    return comm.value, destroy


class DistributedFactorization:
    """Factorization of one pattern on `world` GPUs (this process = `rank` of
    the process group `group`).  world == 1 is the plain single-GPU engine.
    dist_top: None = the engine decides (weight of the top tree), True / False = top tree
    distributed over the ranks / replicated on every rank."""

    def __init__(self, n, ptr, row, nb, rank, world, order=None, nemin=32, panel_width=None,
                 group=None, dist_top=None, engine_flags=0, driver=None, comm=None):
        """driver: "library" -- the exchanges run INSIDE libspllt_hip.so on an ncclComm_t
        (spllt_hip_set_communicator: the path a C / Fortran caller of the unchanged spllt_iface.h
        gets; `comm` = an existing ncclComm_t handle, else one is created over the process group) --
        or "python" -- this module drives them through torch.distributed (the gloo rehearsals and
        CPU tests need that: RCCL has no two ranks on one device).  Default: "library" when the
        process group's backend is nccl, unless SPLLT_MG_DRIVER says otherwise."""
        import torch
        import torch.distributed as dist
        from . import api
        self.rank, self.world, self.group = rank, world, group
        driver_arg = driver
        if driver is None:
            driver = os.environ.get("SPLLT_MG_DRIVER") or (
                "library" if (world > 1 and dist.is_initialized() and dist.get_backend(group) == "nccl") else "python")
        self.driver = driver if world > 1 or comm is not None else "python"
        self._comm_destroy = None
        if dist_top is not None:
            engine_flags |= 8192 if dist_top else 16384
        self.f = api.Factorization(n, ptr, row, nb=nb, nemin=nemin, prune_tree=world > 1,
                                   ncpu=world, order=order, panel_width=panel_width,
                                   engine_flags=engine_flags)
        self.xelems = self.f.set_partition(rank, world) if world > 1 else 0
        self.xbuf = torch.zeros(max(self.xelems, 1), dtype=torch.float64, device="cuda")
        self.ext = None
        self.plan, self.scratch = [], None
        if world > 1:
            self.f.set_exchange_buffer(self.xbuf.data_ptr())
            # the engine's own stream as a torch stream: collectives enqueued under it are
            # ordered behind the pack and in front of the unpack without a host round trip
            self.ext = torch.cuda.ExternalStream(self.f.engine_stream())
            self.plan = exchange_plan(self.f)
            chunk = max([st[2] for st in self.plan] + [0])
            if chunk:
                self.scratch = torch.empty(chunk, dtype=torch.float64, device="cuda")
        self.dist_top = any(st[0] == X_REDUCE_OWNER for st in self.plan)
        if self.driver == "library":
            # the library runs every collective itself, on its own streams.  When the driver was
            # chosen by default (not asked for), a rank that cannot set the communicator up makes
            # EVERY rank fall back to the Python driver -- agreed through the process group, so that
            # nobody is left alone in a collective.
            explicit = driver_arg is not None or bool(os.environ.get("SPLLT_MG_DRIVER"))
            ok, why = 1, ""
            try:
                if comm is None:
                    comm, self._comm_destroy = make_rccl_communicator(rank, world, group)
                self.f.set_communicator(comm)
            except Exception as e:   # noqa: BLE001 - decided collectively below
                if explicit:
                    raise
                ok, why = 0, repr(e)[:200]
            if not explicit and world > 1 and dist.is_initialized():
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                ok = int(flag.item())
            if not ok:
                if rank == 0:
                    print(f"spllt-hip: library RCCL driver not available ({why or 'another rank failed'}): "
                          "falling back to the torch.distributed driver", file=sys.stderr, flush=True)
                try:
                    self.f.set_communicator(0)
                except Exception:   # noqa: BLE001
                    pass
                if self._comm_destroy is not None:
                    self._comm_destroy()
                    self._comm_destroy = None
                self.driver = "python"
        self.stream_ordered = (world > 1 and dist.is_initialized() and
                               (dist.get_backend(group) == "nccl" or
                                bool(os.environ.get("SPLLT_FORCE_STREAM_ORDERED"))))   # (tests: gloo)
        self.phase_ms = {}

    def _exchange(self, k):
        """exchange k of the program: the collective on the engine's stream, between the
        engine's pack and unpack"""
        import torch
        step = self.plan[k]
        # the stream THIS exchange is packed / unpacked on (the per-level reduce-scatters of a
        # distributed top tree run on a side stream of the engine)
        ext = torch.cuda.ExternalStream(self.f.exchange_stream())
        if self.stream_ordered:
            with torch.cuda.stream(ext):
                run_exchange(self.xbuf, step, self.rank, self.world, self.group, self.scratch)   # RCCL: enqueue only
        else:
            # gloo (CPU tests / one-GPU rehearsal) stages through the host: plain syncs
            ext.synchronize()
            run_exchange(self.xbuf, step, self.rank, self.world, self.group, self.scratch)
            torch.cuda.synchronize()

    def factor(self, dval, timed_phases=False):
        """One complete distributed factorization (dval: cuda float64 tensor).  The phases
        -- own subtrees, extend-add, top tree (with its broadcasts when it is distributed) --
        are enqueued back to back on the engine's stream; the host only waits at the end
        (timed_phases=True adds a host synchronisation around the first exchange to time the
        phases).  A pivot failure on any rank raises the same SplltError(-20) on every rank
        after the last phase."""
        t0 = time.perf_counter()
        self.f.factor_dev(dval.data_ptr())
        t1 = t2 = t0
        nx = 0
        if self.driver == "library":
            # (every exchange was issued inside spllt_hip_factor_dev; the phases are not separable here)
            self.f.wait()
            t3 = time.perf_counter()
            self.phase_ms = {"total": (t3 - t0) * 1e3, "driver": "library (spllt_hip_set_communicator)"}
            return self
        while True:
            k = self.f.pending_exchange()
            if k < 0:
                break
            if timed_phases and nx == 0:
                self.ext.synchronize()
                t1 = time.perf_counter()
            self._exchange(k)
            if timed_phases and nx == 0:
                self.ext.synchronize()
                t2 = time.perf_counter()
            self.f.continue_after_exchange()
            nx += 1
        self.f.wait()
        t3 = time.perf_counter()
        if timed_phases or self.world == 1:
            self.phase_ms = {"subtrees": (t1 - t0) * 1e3, "exchange": (t2 - t1) * 1e3,
                             "top": (t3 - t2) * 1e3}
        return self

    def owned_mask(self):
        """bool per pivot position: True where this rank contributes the entry of a
        distributed vector (its own subtrees; rank 0 also the top tree)"""
        if self.world == 1:
            return np.ones(self.f.n, dtype=bool)
        owner = self.f.partition("owner")
        sptr = self.f.sym("sptr")
        node_of = np.repeat(np.arange(len(sptr) - 1), np.diff(sptr))
        own = owner[node_of]
        return (own == self.rank) | ((own < 0) & (self.rank == 0))

    def solve(self, b):
        """Distributed solve A x = b (b: numpy, n or n x nrhs; every rank passes the
        same b and gets the same x).  Forward substitution on the own subtrees, one
        all-reduce of the right-hand side vector (n doubles per rhs), the top tree
        on every rank, backward substitution on the own subtrees, one all-reduce
        to assemble x.  Mirrors the exchange of the factorization (extend-add of
        the reference's solve tasks, src/spllt_solve_mod.F90)."""
        import torch
        import torch.distributed as dist
        b = np.asarray(b, dtype=np.float64)
        one = b.ndim == 1
        B = b.reshape(self.f.n, -1, order="F")
        nrhs = B.shape[1]
        pos = self.f.sym("order")                      # 0-based pivot position of variable i
        mask = torch.tensor(self.owned_mask(), device="cuda")
        Y = np.zeros((nrhs, self.f.n))
        Y[:, pos] = B.T
        y = torch.tensor(Y, device="cuda")             # y[q, p]: pivot order, rhs-major
        y *= mask
        # The engine launches on its own (non-blocking) HIP stream and solve_dev only
        # synchronises that one; torch's elementwise ops and the RCCL all-reduce run on
        # torch's streams.  Every hand-over between the two is a full device sync.
        torch.cuda.synchronize()
        if self.world == 1:
            self.f.solve_dev(y.data_ptr(), nrhs, 0, -1)
        else:
            self.f.solve_dev(y.data_ptr(), nrhs, 0, 0)
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize()
            self.f.solve_dev(y.data_ptr(), nrhs, 0, 1)
            self.f.solve_dev(y.data_ptr(), nrhs, 0, 2)
            y *= mask
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize()
        X = y.cpu().numpy()[:, pos].T
        return X[:, 0].copy() if one else np.asfortranarray(X)

    def close(self):
        self.f.close()
        if self._comm_destroy is not None:
            self._comm_destroy()
            self._comm_destroy = None


def _timed(df, dval, steps, active):
    """max-over-ranks wall time of `steps` factorizations by the active ranks
    (idle ranks only take part in the barriers)."""
    import torch
    import torch.distributed as dist
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if active:
        for _ in range(steps):
            df.factor(dval)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt.item())


def bench_distributed(args, A, n, ptr, row, val, order, nb, name, rank, world, roofline_fn=None):
    """bench.py body for N > 1 (strong scaling: one factorization, N GPUs).

    `value` is ALWAYS the run that uses all N ranks (partition width = N).  The
    tree-level partition could also be run narrower (w < N ranks own subtrees, the
    others idle: a wider partition shortens the subtree phase but grows the replicated
    top tree and the exchange); with SPLLT_WIDTH_SWEEP=1 the narrower widths are timed
    too and reported under detail.width_trials_ms -- never as `value`."""
    import torch
    import torch.distributed as dist
    dval = torch.tensor(val, device="cuda")
    trial = {}
    if os.environ.get("SPLLT_WIDTH_SWEEP"):
        for w in sorted({w for w in (1, 2, 4, 8) if w < world}):
            grp = dist.new_group(ranks=list(range(w))) if w > 1 else None
            active = rank < w
            df = None
            if active:
                df = DistributedFactorization(n, ptr, row, nb, rank, w, order=order,
                                              panel_width=args.panel, group=grp)
            _timed(df, dval, 1, active)                      # first touch
            t = _timed(df, dval, max(1, args.warmup), active) / max(1, args.warmup)
            trial[w] = round(t * 1e3, 3)
            if df is not None:
                df.close()
            torch.cuda.empty_cache()
    w = world
    df = DistributedFactorization(n, ptr, row, nb, rank, w, order=order, panel_width=args.panel)
    for _ in range(max(1, args.warmup)):
        _timed(df, dval, 1, True)
    t_total = _timed(df, dval, args.steps, True)
    trial[w] = round(t_total / args.steps * 1e3, 3)
    df.factor(dval, timed_phases=True)       # one more, untimed for `value`: per-phase times
    si_t = torch.zeros(4, dtype=torch.float64, device="cuda")
    if rank == 0:
        si = df.f.sym_info()
        si_t = torch.tensor([si["flops"], si["nnz_l"], si["nnodes"], df.xelems], dtype=torch.float64,
                            device="cuda")
    flops = float(si_t[0].item()) if rank == 0 else 0.0
    check, own_w, top_flops = {}, np.zeros(max(w, 1)), 0.0
    if w > 1:
        # flops of the branches each rank owns = subtree weights of their roots
        owner, wgt, par = df.f.partition("owner"), df.f.sym("weight"), df.f.sym("sparent")
        nn = len(owner)
        for s in range(nn):
            if owner[s] >= 0 and (par[s] >= nn or owner[par[s]] < 0):
                own_w[owner[s]] += wgt[s]
        top_flops = float(df.f.sym_info()["flops"]) - own_w.sum()
    if not args.no_check:
        check = _accuracy_gate(df, A, n, rank, w, True)
    # roofline of the dominant kernel: the partitioned program has no per-launch profile (its
    # exchange points are collectives), so rank 0 measures the single-GPU program of the SAME
    # workload, live, after the timed region (the other ranks wait at the next collective)
    roof = None
    if rank == 0 and roofline_fn is not None:
        try:
            from . import api
            f1 = api.Factorization(n, ptr, row, nb=nb, nemin=getattr(args, "nemin", 32), prune_tree=False, order=order,
                                   panel_width=args.panel)
            f1.factor_dev(dval.data_ptr()).wait()
            roof = roofline_fn(f1, val)
            roof["note"] = ("measured on rank 0 with the single-GPU program of the same workload (the partitioned "
                            "program's exchange points are collectives: no per-launch profile)")
            f1.close()
        except Exception as e:   # noqa: BLE001 - a diagnostic, never a reason to lose the line
            roof = {"error": repr(e)[:200]}
    out = None
    if rank == 0:
        out = {
            "metric": "factorize GFLOP/s (fp64)",
            "value": round(flops / (t_total / args.steps) / 1e9, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_total / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": name, "n": n, "nb": nb, "nnz_L": int(si_t[1].item()),
                       "flops_sym": flops, "nnodes": int(si_t[2].item()),
                       "parallelism": (f"subtree partition over all {world} GPUs + RCCL reduce-scatter (extend-add) "
                                       "to the owners of a 1-D block-column-cyclic top tree, one broadcast "
                                       "per finished block-column step" if df.dist_top else
                                       f"subtree partition over all {world} GPUs + one RCCL all-reduce "
                                       "(extend-add) on the engine's stream, replicated top tree")},
            "roofline": roof, "cpu_baseline": None,
            "detail": {"partition_width": w, "width_trials_ms": trial, "distributed_top_tree": df.dist_top,
                       "exchange_driver": df.driver,
                       "exchanges": len(df.plan),
                       "phase_ms_rank0": df.phase_ms, "exchange_MB": df.xelems * 8 / 1e6,
                       "subtree_gflop_per_rank": (own_w / 1e9).round(1).tolist(),
                       "top_tree_gflop": round(top_flops / 1e9, 1), "check": check,
                       "baseline_config_for_this_n": None},
        }
    # The headline measurement is complete.  The configuration BASELINE.json names for this GPU
    # count is a second, larger distributed factorization: it must not be able to take the
    # headline with it.  A failure, or no answer within SPLLT_EXTRA_TIMEOUT_S (default 420 s) --
    # a rank stuck in a collective cannot be interrupted from Python -- makes rank 0 print the
    # JSON line with the error noted and every rank leave the process.
    if not os.environ.get("SPLLT_NO_BASELINE_CONFIG"):
        import json
        import threading
        done = threading.Event()

        def leave(why):
            if rank == 0:
                out["detail"]["baseline_config_for_this_n"] = {"error": why}
                print(json.dumps(out), flush=True)
            os._exit(0)

        def guard():
            if not done.wait(float(os.environ.get("SPLLT_EXTRA_TIMEOUT_S", "420"))):
                leave("no answer in time: abandoned")
        threading.Thread(target=guard, daemon=True).start()
        try:
            extra = _baseline_config_for(world, args, rank)
        except BaseException as e:   # noqa: BLE001 - the other ranks follow through their guards
            leave(repr(e)[:300])
        done.set()
        if rank == 0:
            out["detail"]["baseline_config_for_this_n"] = extra
    return out


# BASELINE.json names a configuration per GPU count: Flan_1565 on 2 and 4, Serena on 8
BASELINE_CONFIG_FOR_N = {2: "flan_like", 4: "flan_like", 8: "serena_like"}


def _baseline_config_for(world, args, rank):
    """Times the configuration BASELINE.json quotes for this GPU count (stand-in matrices)
    with the same engine: reported under detail, never as `value`."""
    import torch
    from . import api, matgen
    cfg_name = BASELINE_CONFIG_FOR_N.get(world)
    if cfg_name is None:
        return None
    A, order, cfg = matgen.build_config(cfg_name, 1.0)
    n, ptr, row, val = api.csc_lower_1based(A)
    dval = torch.tensor(val, device="cuda")
    df = DistributedFactorization(n, ptr, row, cfg["nb"], rank, world, order=order)
    _timed(df, dval, 1, True)
    steps = 2
    t = _timed(df, dval, steps, True) / steps
    df.factor(dval, timed_phases=True)
    si = df.f.sym_info()
    out = {"workload": cfg_name, "n": n, "nb": cfg["nb"], "flops_sym": float(si["flops"]),
           "ms_per_step": round(t * 1e3, 2), "gflops": round(float(si["flops"]) / t / 1e9, 1),
           "phase_ms_rank0": df.phase_ms, "exchange_MB": df.xelems * 8 / 1e6,
           "distributed_top_tree": df.dist_top, "exchanges": len(df.plan)}
    df.close()
    del dval
    torch.cuda.empty_cache()
    return out


def _accuracy_gate(df, A, n, rank, w, active):
    """Distributed solve of A x = A 1 on the active ranks, residuals on rank 0.
    Never raises and never leaves a rank behind in a collective."""
    import torch
    import torch.distributed as dist
    check, ok, x = {}, 1, None
    b = A @ np.ones(n)
    if active:
        try:
            x = df.solve(b)
        except Exception as e:   # a failing rank makes the others fail in the collective too
            ok, check = 0, {"error": repr(e)[:200]}
    okt = torch.tensor([ok], dtype=torch.int32, device="cuda")
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if int(okt.item()) == 1 and rank == 0:
        r = b - A @ x
        check = {"resid_2norm_rel": float(np.linalg.norm(r) / np.linalg.norm(b)),
                 "bwd_err": float(np.linalg.norm(r) /
                                  (np.linalg.norm(b) + abs(A).max() * np.linalg.norm(x))),
                 "solve": "distributed (spllt_hip_solve_dev phases + 2 all-reduces of the rhs vector)"}
    return check
