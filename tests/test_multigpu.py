"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the subtree partition,
the EXCHANGE step (spllt_amd.multigpu.reduce_exchange_buffer == the production
all-reduce) and the two-phase program, with the program tables interpreted in
numpy (tests/emulate.py) instead of executed by HIP."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case, top, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emulate import emulate_program
        from helpers import dense_arena, lower_mask, make_case
        from spllt_amd import matgen, multigpu
        A = {"p2d": lambda: matgen.poisson2d(28), "box": lambda: matgen.nd_like((9, 8, 8), 2),
             "p3d": lambda: matgen.poisson3d(9)}[case]()
        f, val = make_case(A, nb=16, nemin=8, prune=True, ncpu=world, panel_width=16,
                           engine_flags={"replicated": 16384, "distributed": 8192}[top])
        xel = f.set_partition(rank, world)
        owner = f.partition("owner")
        plan = multigpu.exchange_plan(f)

        def exchange(k, xbuf):
            assert xbuf.size == xel
            t = torch.from_numpy(xbuf.copy())
            multigpu.run_exchange(t, plan[k], rank, world)      # the production collective, over gloo
            return t.numpy()

        got = emulate_program(f, val, exchange=exchange, partitioned=True)
        exp = dense_arena(f, A)
        mask = lower_mask(f)
        # this rank must hold its own subtrees and the whole top tree
        bc_node = f.sym("bcol_node")
        off, w, nr = f.sym("bcol_off"), f.sym("bcol_width"), f.sym("bcol_nrow")
        mine = np.zeros_like(mask)
        for b in range(len(off)):
            o = owner[bc_node[b]]
            if o == rank or o < 0:
                mine[off[b]:off[b] + nr[b] * w[b]] = True
        err = np.abs(got - exp)[mask & mine].max() / np.abs(exp).max()
        untouched = np.all(got[mask & ~mine] == 0.0)
        nsub = int((f.sym("small") == 1).sum())
        ntop = int((owner < 0).sum())
        L = f.program("launches")
        ret[rank] = (float(err), bool(untouched), nsub, ntop, int((L[:, 0] == 2).sum()),
                     sorted(set(owner.tolist())), [p[0] for p in plan])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("top", ["replicated", "distributed"])
@pytest.mark.parametrize("case,world", [("p2d", 2), ("box", 2), ("p3d", 3)])
def test_partitioned_program_over_gloo(case, world, top):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + world + (10 if top == "distributed" else 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, case, top, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        err, untouched, nsub, ntop, nx, owners, kinds = ret[rank]
        assert err < 1e-13, (rank, err)
        assert untouched            # other ranks' subtrees are never written here
        assert nsub >= world and ntop >= 1
        if top == "replicated":
            assert nx == 1 and kinds == [0]          # exactly one exchange point: the all-reduce
        else:
            # reduce-scatter to the owners (one per level of the top tree), a broadcast per finished
            # block-column step, the indicator
            nred = kinds.count(1)
            assert nx == len(kinds) >= 3 and nred >= 1 and kinds[:nred] == [1] * nred
            assert kinds[-1] == 3 and set(kinds[nred:-1]) == {2}
        assert set(owners) == set(range(world)) | {-1}


def test_owner_assignment_is_balanced_and_covers_subtrees():
    """proportional mapping: owned territories are closed downward (whole
    branches, separators included), the top tree (-1) is closed upward, and the
    ranks' loads are balanced"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_case
    from spllt_amd import matgen
    A = matgen.poisson2d(48)
    for world in (2, 3, 4):
        f, _ = make_case(A, nb=32, nemin=16, prune=True, ncpu=world)
        f.set_partition(0, world)
        owner, w, sparent = f.partition("owner"), f.sym("weight"), f.sym("sparent")
        nn = len(owner)
        load = np.zeros(world)
        for s in range(nn):
            assert -1 <= owner[s] < world
            p = sparent[s]
            if p < nn:
                if owner[p] >= 0:
                    assert owner[s] == owner[p]      # below an owned node everything is owned by it
                elif owner[s] >= 0:
                    load[owner[s]] += w[s]           # root of an owned branch
            elif owner[s] >= 0:
                load[owner[s]] += w[s]
        assert (owner == -1).any() and load.min() > 0.25 * load.max(), (world, load)
        # the top tree is what spans several ranks: far fewer nodes than the branches
        assert (owner == -1).sum() < 0.25 * nn


def _gpu_worker(rank, world, port, ret, stream_ordered=False, dist_top=False):
    """one process per rank, every rank on device 0, reduction over gloo: the production
    DistributedFactorization (factor with the exchange on the engine's stream, then the
    three-phase solve) end to end"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if stream_ordered:
        # the RCCL code path (collective enqueued under the engine's stream wrapped as a torch
        # ExternalStream, no host synchronisation of ours between the phases), driven through gloo
        os.environ["SPLLT_FORCE_STREAM_ORDERED"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spllt_amd import api, matgen, multigpu
        A = matgen.nd_like((11, 10, 9), 2)
        n, ptr, row, val = api.csc_lower_1based(A)
        df = multigpu.DistributedFactorization(n, ptr, row, 48, rank, world, nemin=16, dist_top=dist_top)
        assert df.dist_top == dist_top and len(df.plan) == (1 if not dist_top else len(df.plan)) >= 1
        dval = torch.tensor(val, device="cuda")
        torch.cuda.synchronize()
        df.factor(dval)
        df.factor(dval, timed_phases=True)          # re-factorization, with per-phase syncs
        rng = np.random.default_rng(11)
        X = rng.standard_normal((n, 3))
        B = A @ X
        got = df.solve(B)
        r = B - A @ got
        bwd = max(float(np.linalg.norm(r[:, q]) / (np.linalg.norm(B[:, q]) + abs(A).max() * np.linalg.norm(got[:, q])))
                  for q in range(3))
        ret[rank] = (bwd, float(np.abs(got - X).max()), sorted(df.phase_ms))
        df.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dist_top", [False, True])
@pytest.mark.parametrize("stream_ordered", [False, True])
def test_distributed_factorization_two_processes_one_gpu(stream_ordered, dist_top):
    """spllt_amd.multigpu.DistributedFactorization in two processes that share the one GPU, over
    gloo: top tree replicated (one all-reduce) or distributed (reduce-scatter to the owners, a
    broadcast per block-column step); the collectives ordered by host syncs or enqueued under
    the engine's stream (the RCCL code path)."""
    import torch.multiprocessing as mp
    world = 2
    port = 31500 + (os.getpid() % 2000) + (7 if stream_ordered else 0) + (14 if dist_top else 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, port, ret, stream_ordered, dist_top), nprocs=world, join=True)
    assert len(ret) == world
    for rank in range(world):
        bwd, err, phases = ret[rank]
        assert bwd <= 1e-14 and err <= 1e-9, (rank, bwd, err)
        assert phases == ["exchange", "subtrees", "top"]


def _rccl_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from spllt_amd import multigpu
        st = torch.cuda.Stream()
        ext = torch.cuda.ExternalStream(st.cuda_stream)      # like the engine's stream in production
        x = torch.arange(1000, dtype=torch.float64, device="cuda")
        ref = x.clone()
        scratch = torch.empty(600, dtype=torch.float64, device="cuda")
        plan = [(multigpu.X_REDUCE_OWNER, 600, 600, []),          # one rank: its chunk is everything
                (multigpu.X_BCAST, 900, 0, [(0, 0, 300), (0, 300, 600)]),
                (multigpu.X_REDUCE_ALL, 1000, 0, []), (multigpu.X_FLAG, 1, 0, [])]
        with torch.cuda.stream(ext):
            for step in plan:
                multigpu.run_exchange(x, step, 0, 1, None, scratch, force=True)
        st.synchronize()
        ret[0] = bool(torch.equal(x, ref))      # sums / broadcasts over one rank change nothing
    finally:
        dist.destroy_process_group()


def _rccl_inlib_worker(rank, world, port, ret, flags):
    """one process, one GPU, a ONE-rank RCCL communicator created with the process's own librccl
    (ncclGetUniqueId + ncclCommInitRank through ctypes: no torch.distributed involved)"""
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.cuda.set_device(0)
    os.environ["SPLLT_HIP_COMM_REHEARSAL"] = "1"
    from helpers import lower_mask, make_case, rel_err
    from spllt_amd import matgen
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        A = matgen.nd_like((12, 11, 10), 2)
        # (a) the library drives the exchanges itself on the communicator
        f, val = make_case(A, nb=64, nemin=16, prune=True, ncpu=2, engine_flags=flags)
        f.set_partition(0, 2)
        f.set_communicator(comm.value)
        nx = len(f.program("exchanges"))
        got = f.factor(val).wait().get_factor()
        assert f.pending_exchange() < 0
        b = A @ np.ones(f.n)
        f.solve(b)                                   # (two all-reduces inside; one rank: its own share only)
        # (b) the caller drives them (spllt_hip_pending_exchange / spllt_hip_continue) and does
        # what collectives over ONE rank do to the buffer: nothing
        g, _ = make_case(A, nb=64, nemin=16, prune=True, ncpu=2, engine_flags=flags)
        xe = g.set_partition(0, 2)
        xbuf = torch.zeros(max(xe, 1), dtype=torch.float64, device="cuda")
        g.set_exchange_buffer(xbuf.data_ptr())
        dval = torch.tensor(val, device="cuda")
        torch.cuda.synchronize()
        g.factor_dev(dval.data_ptr())
        n2 = 0
        while g.pending_exchange() >= 0:
            g.wait()
            g.continue_after_exchange()
            n2 += 1
        ref = g.wait().get_factor()
        ret[0] = (nx, n2, float(rel_err(got, ref, lower_mask(f))), bool(np.isfinite(got).all()))
        f.close()
        g.close()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
@pytest.mark.parametrize("top", ["replicated", "distributed"])
def test_exchanges_inside_the_library_on_rccl_single_rank(top):
    """spllt_hip_set_communicator: with the caller's ncclComm_t the library runs the exchange loop
    of the partition itself -- ncclAllReduce / in-place ncclReduceScatter / grouped ncclBroadcast
    on the engine's stream between pack and unpack -- so that spllt_factor / spllt_wait /
    spllt_solve of the unchanged C-ABI are all a one-process-per-GPU caller needs (the reference's
    distributed build: src/PaRSEC/spllt_parsec_blk_data.c:33-64).  This box has one GPU: rank 0 of
    a 2-rank partition on a ONE-rank communicator (rehearsal switch), against the same program with
    the caller driving the exchanges; the multi-rank semantics are covered over gloo."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    flags = 16384 if top == "replicated" else 8192
    mp.spawn(_rccl_inlib_worker, args=(1, 0, ret, flags), nprocs=1, join=True)
    nx, n2, err, finite = ret[0]
    assert nx == n2 and nx >= (1 if top == "replicated" else 3)
    assert finite and err <= 1e-12, ret[0]


def _library_driver_worker(rank, world, port, ret):
    """one process, one GPU, a ONE-rank nccl process group: DistributedFactorization with the
    library driver (the path bench.py --gpus N takes on a real node) creates the ncclComm_t itself
    -- unique id from rank 0, shipped over the process group -- and hands it to
    spllt_hip_set_communicator; here the one-rank communicator stands in for rank 0 of 2"""
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.cuda.set_device(0)
    os.environ["SPLLT_HIP_COMM_REHEARSAL"] = "1"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from spllt_amd import api, matgen, multigpu
        A = matgen.nd_like((12, 11, 10), 2)
        n, ptr, row, val = api.csc_lower_1based(A)
        comm, destroy = multigpu.make_rccl_communicator(0, 1)
        dval = torch.tensor(val, device="cuda")
        out = {}
        for drv in ("library", "python"):
            # width-2 partition, this process is rank 0 of it; library: exchanges inside factor_dev
            df = multigpu.DistributedFactorization(n, ptr, row, 64, 0, 2, nemin=16, dist_top=True, driver=drv,
                                                   comm=comm if drv == "library" else None)
            assert df.driver == drv
            df.factor(dval)
            out[drv] = df.f.get_factor()
            out[drv + "_nx"] = len(df.plan)
            if drv == "library":
                assert df.f.pending_exchange() < 0 and "library" in df.phase_ms.get("driver", "")
                assert "rehearsal" in df.f.last_error()
            df.close()
        destroy()
        a, b = np.nan_to_num(out["library"]), np.nan_to_num(out["python"])
        same = float(np.abs(a - b).max() / np.abs(b).max()) <= 1e-12     # (fp64 atomics: equal to rounding)
        ret[0] = (bool(same), out["library_nx"], bool(np.isfinite(out["library"]).all()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_path_library_driver_on_rccl_single_rank():
    """multigpu.DistributedFactorization(driver="library") -- what bench.py --gpus N uses when the
    backend is nccl: ncclGetUniqueId on rank 0, the id shipped through the process group,
    ncclCommInitRank, spllt_hip_set_communicator, and then nothing but factor / wait -- against the
    Python driver of the same exchanges (torch.distributed on the engine's stream), with the one
    rank this box has (a one-rank communicator and group standing in for rank 0 of a width-2
    partition: both drivers leave the other rank's parts out in the same way)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_library_driver_worker, args=(1, 33700 + (os.getpid() % 2000), ret), nprocs=1, join=True)
    same, nx, finite = ret[0]
    assert finite and nx >= 3 and same, ret[0]


@pytest.mark.gpu
def test_run_exchange_calls_on_rccl_single_rank():
    """The collectives of multigpu.run_exchange (all-reduce, reduce_scatter_tensor into the
    scratch + copy to the rank's chunk, broadcasts of buffer slices) issued on a real RCCL
    process group under an ExternalStream, as in production -- with the one rank this box has.
    Checks the call signatures / views RCCL accepts; the multi-rank semantics are covered over
    gloo."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, 33500 + (os.getpid() % 2000), ret), nprocs=1, join=True)
    assert ret.get(0) is True


@pytest.mark.gpu
@pytest.mark.parametrize("dist_top", ["0", "1"])
def test_bench_contract_two_ranks_over_gloo(dist_top):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per
    process), rehearsed on the one-GPU box: both ranks on device 0, gloo instead of RCCL.  Rank 0
    must print ONE JSON line with the contract's fields, `value` from the width-2 run, and a
    residual that passes; with the top tree replicated and distributed."""
    import json
    import subprocess
    port = 34500 + (os.getpid() % 2000) + (5 if dist_top == "1" else 0)
    env = dict(os.environ, SPLLT_DIST_BACKEND="gloo", SPLLT_SINGLE_DEVICE="1", SPLLT_NO_BASELINE_CONFIG="1",
               SPLLT_DIST_TOP=dist_top)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--scale", "0.5"], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["detail"]["partition_width"] == 2
    assert d["detail"]["distributed_top_tree"] == (dist_top == "1")
    assert d["detail"]["check"]["bwd_err"] <= 1e-14
    # the roofline object of the N > 1 line: rank 0, live, on the single-GPU program of the same workload
    roof = d["roofline"]
    assert roof and "error" not in roof, roof
    assert roof["bound"] == "mfma" and roof["peak"] == 78.6 and 0 < roof["frac"] <= 1 and "single-GPU program" in roof["note"]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_top_tree_block_column_owners(world):
    """Distributed top tree: every block column of the top tree has exactly one owner, the owners
    are dealt round-robin in the order the top tree is walked (level, block-column step, node) --
    so the block columns of one step sit on different ranks as far as there are ranks -- and
    nothing outside the top tree has one.  The exchanges of every rank's program are the same list."""
    from helpers import make_case
    from spllt_amd import matgen
    A = matgen.nd_like((14, 13, 12), 2)
    fs = []
    for r in range(world):
        f, _ = make_case(A, nb=32, nemin=8, prune=True, ncpu=world, engine_flags=8192)
        f.set_partition(r, world)
        fs.append(f)
    f = fs[0]
    owner, towner = f.partition("owner"), f.partition("top_bcol_owner")
    bc_node, level, nb0 = f.sym("bcol_node"), f.sym("level"), f.sym("node_bcol0")
    top = np.nonzero(owner[bc_node] < 0)[0]
    assert len(towner) == len(bc_node)
    assert np.all(towner[top] >= 0) and np.all(towner[top] < world)
    assert np.all(np.delete(towner, top) == -1)
    # walk order: level, step c, node
    order = sorted(top.tolist(), key=lambda b: (level[bc_node[b]], b - nb0[bc_node[b]], bc_node[b]))
    assert [int(towner[b]) for b in order] == [i % world for i in range(len(order))]
    if len(top) >= world:
        assert set(towner[top].tolist()) == set(range(world))
    ex0, it0 = f.program("exchanges"), f.program("xitems")
    # the extend-add first -- one reduce-scatter per level of the top tree, lowest level first, each
    # with a region of its own in the buffer --, then the broadcasts of the steps, the flag last
    nred = int((ex0[:, 0] == 1).sum())
    assert nred == len(set(level[bc_node[top]].tolist())) >= 1
    assert np.all(ex0[:nred, 0] == 1) and ex0[-1, 0] == 3 and np.all(ex0[nred:-1, 0] == 2)
    red_levels = [sorted(set(level[bc_node[it0[first:first + n, 0]]].tolist())) for kind, first, n, *_ in ex0[:nred].tolist()]
    assert all(len(lv) == 1 for lv in red_levels) and [lv[0] for lv in red_levels] == sorted(lv[0] for lv in red_levels)
    ends = ex0[:nred, 3].tolist()
    assert ends == sorted(ends) and all(e - world * c == (ends[i - 1] if i else 0) for i, (e, c) in enumerate(zip(ends, ex0[:nred, 4].tolist())))
    bcast_lo = min([int(it0[first:first + n, 2].min()) for kind, first, n, *_ in ex0.tolist() if kind == 2 and n] + [ends[-1]])
    assert bcast_lo >= ends[-1], "the broadcasts use the buffer behind the reduce regions"
    for g in fs[1:]:
        assert np.array_equal(g.program("exchanges"), ex0) and np.array_equal(g.program("xitems"), it0)
        assert np.array_equal(g.partition("top_bcol_owner"), towner)
    # a broadcast item's root is the owner; the reduce-scatter chunks are disjoint and in rank order
    for kind, first, n, elems, chunk in ex0.tolist():
        items = it0[first:first + n]
        if kind == 2:
            assert all(int(towner[b]) == root for b, root in items[:, :2].tolist())
        if kind == 1:
            base = elems - chunk * world
            for b, root, xo, cnt, off, space in items.tolist():
                assert base + root * chunk <= xo and xo + cnt <= base + (root + 1) * chunk and space == 0


def test_top_tree_is_distributed_only_when_it_pays():
    """The engine's choice (distribute_top_tree): a top tree that is a chain of dependent panel
    steps (small problem) stays replicated -- distributing it would only add a broadcast per
    step --, a compute-bound one is distributed; flags 8192 / 16384 force either."""
    from helpers import make_case
    from spllt_amd import matgen
    small = matgen.nd_like((10, 9, 8), 2)
    f, _ = make_case(small, nb=32, nemin=8, prune=True, ncpu=4)
    f.set_partition(0, 4)
    assert len(f.partition("top_bcol_owner")) == 0            # replicated by choice
    for flag, want in ((8192, True), (16384, False)):
        g, _ = make_case(small, nb=32, nemin=8, prune=True, ncpu=4, engine_flags=flag)
        g.set_partition(0, 4)
        assert (len(g.partition("top_bcol_owner")) > 0) == want
    big = matgen.poisson3d(112)                                # 5.5 TFLOP, most of it in the top of the tree
    h, _ = make_case(big, nb=384, nemin=32, prune=True, ncpu=4)
    h.set_partition(0, 4)
    assert len(h.partition("top_bcol_owner")) > 0              # distributed by choice


def test_extend_add_is_pipelined_by_top_tree_level():
    """Distributed top tree, multi-stream program: the reduce-scatter of the extend-add is one
    exchange per LEVEL of the top tree, lowest first, on the side stream ("pipeline per
    ancestor node so the reduce overlaps"; the reference's walk is per destination tile,
    src/spllt_factorization_mod.F90:39-191) -- the panel chains of top level l wait for chunk l
    only, the updates that land in a level wait for that level's chunk (the unpack overwrites),
    nothing waits for more than it needs.  Checked: the structure, the stream DAG of every rank
    (every conflict ordered), the numbers in lockstep -- and that the order really hangs on the
    per-level waits (mutation: a rank whose program lost the wait in front of a top level is caught
    by the DAG check)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_schedule as ts
    from emulate import emulate_ranks
    from helpers import dense_arena, lower_mask, make_case
    from spllt_amd import matgen
    A = matgen.nd_like((12, 11, 10), 2)
    world = 4
    fs = []
    for r in range(world):
        f, val = make_case(A, nb=32, nemin=8, prune=True, ncpu=world, panel_width=16, engine_flags=8192)
        f.set_partition(r, world)
        f.rank = r
        fs.append(f)
    f = fs[0]
    L, ex = f.program("launches"), f.program("exchanges")
    xl = L[L[:, 0] == 2]
    nred = int((ex[:, 0] == 1).sum())
    assert nred >= 2, "the case is meant to have a top tree of several levels"
    assert (xl[:nred, 6] == 3).all(), "the per-level reduce-scatters run on the side stream"
    assert (xl[nred:, 6] == 0).all(), "broadcasts and the flag stay on the chain stream"
    assert xl[0, 8] >= 0 and (xl[1:nred, 8:12] == -1).all(), "the first waits for phase 1, the others follow in their stream"
    recs = xl[:nred, 7].tolist()
    # every chunk's event is waited for by a marker of the chain stream (the level's own start) ...
    waited = {int(w) for row in L[(L[:, 0] == 1) & (L[:, 3] == 0)] for w in row[8:12] if w >= 0}
    assert set(recs) <= waited | {int(w) for row in L for w in row[8:12] if w >= 0}
    for g in fs:
        assert not ts.dag_violations(g)[0]
    arenas = emulate_ranks(fs, val)
    ref, mask = dense_arena(f, A), lower_mask(f)
    owner, bc_node = f.partition("owner"), f.sym("bcol_node")
    off, w, nr = f.sym("bcol_off"), f.sym("bcol_width"), f.sym("bcol_nrow")
    for r, g in enumerate(fs):
        mine = np.zeros_like(mask)
        for b in range(len(off)):
            if owner[bc_node[b]] in (r, -1):
                mine[off[b]:off[b] + nr[b] * w[b]] = True
        assert np.abs(arenas[r] - ref)[mask & mine].max() / np.abs(ref).max() < 1e-12
    # mutation: drop the waits for the LAST chunk from a rank's program -> some launch touches the
    # top level's block columns unordered against the unpack of its reduce-scatter
    class Mutant:
        def __init__(self, g, drop):
            self.g, self.drop, self.rank = g, drop, g.rank
        def program(self, name):
            t = self.g.program(name)
            if name == "launches":
                t = t.copy()
                t[:, 8:12][t[:, 8:12] == self.drop] = -1
            return t
        def sym(self, name):
            return self.g.sym(name)
    assert any(ts.dag_violations(Mutant(g, recs[-1]))[0] for g in fs), "the DAG check must notice a missing chunk wait"
    for g in fs:
        g.close()
