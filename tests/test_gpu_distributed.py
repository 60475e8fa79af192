"""Two ranks sharing the one GPU of the test box (gloo between the processes):
the real HIP phases with row-sharded X, in the replicated mode (all-reduce of U^T U
and of Q, fused single-rank V update on every rank) and in the column-sharded mode
(reduce-scatter / all-gather), must reproduce the single-rank result."""
import os
import socket

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _collect(procs, q, deadline_s=300):
    """Results of the worker processes, drained BEFORE joining (they are larger than a pipe buffer); a worker that
    dies or a run that exceeds the deadline fails the test instead of blocking it."""
    import time
    res = {}
    t0 = time.time()
    while len(res) < len(procs):
        if not q.empty():
            item = q.get()
            res[item[0]] = item[1:]
            continue
        dead = [p for p in procs if p.exitcode not in (None, 0)]
        if dead or time.time() - t0 > deadline_s:
            for p in procs:
                if p.is_alive():
                    p.kill()
            raise AssertionError("worker failed (exit codes %s) or timed out" % [p.exitcode for p in procs])
        time.sleep(0.05)
    for p in procs:
        p.join(60)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    return res


def _worker(rank, world, port, k, x_dtype, mode, q, eager=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prmf_amd import PathwayLibrary
        from prmf_amd.distributed import DistributedPrmf
        from prmf_amd.solver import PrmfEngine
        from prmf_amd.synthetic import shard_rows
        case = synth.small_case(seed=13, m=700, n=1500, k=k, P=6, order=80, bf16_x=(x_dtype == "bf16"))
        X = case["X"]
        lo, hi = shard_rows(X.shape[0], world, rank)
        eng = PrmfEngine(hi - lo, X.shape[1], k, x_dtype=x_dtype, device=0, nranks=world, rank=rank, dist_mode=mode,
                         eager_objective=eager)
        assert eng.dist_mode == mode
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(k) % 6)
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        objs, gd = DistributedPrmf(eng).step(U, V, 6, case["gamma"], case["delta"])
        q.put((rank, U.cpu().numpy(), V.cpu().numpy(), objs))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,x_dtype,mode", [(16, "f32", "replicated"), (13, "bf16", "replicated"), (64, "bf16", "replicated"),
                                             (128, "bf16", "replicated"), (100, "f32", "colshard"), (64, "f32", "colshard"), (48, "f32", "replicated"),
                                             (16, "f32", "colshard"), (13, "bf16", "colshard")])
def test_two_ranks_match_single_rank(k, x_dtype, mode):
    import torch
    import torch.multiprocessing as mp
    from oracle import prmf_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, x_dtype, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]      # replicated state agrees bitwise
    U = np.concatenate([res[0][0], res[1][0]])
    case = synth.small_case(seed=13, m=700, n=1500, k=k, P=6, order=80, bf16_x=(x_dtype == "bf16"))
    pws = [orc.pathway_from_edges(1500, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], [pws[i % 6] for i in range(k)],
                                             n_steps=6, gamma=case["gamma"], delta=case["delta"], trace=tr)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(U, Ur) < 1e-4 and rel(res[0][1], Vr) < 1e-4
    for a, b in zip(res[0][2], tr):
        for key in ("recon", "manifold", "ignore", "fro", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key


@pytest.mark.parametrize("k,x_dtype", [(64, "bf16"), (16, "f32")])
def test_column_sharded_deferred_records_equal_eager(k, x_dtype):
    """Column-sharded mode: the records evaluated at prmf_end from the kept Grams, the gathered per-workgroup sums and
    V_new on the supports (V^T V and the sums reduced by the next sweep's prologue) against the per-iteration form
    (PRMF_FLAG_EAGER_OBJECTIVE): the factors bit for bit, the records to the last bits (the per-iteration form adds
    manifold / ignore per rank first, the deferred one over all k columns at once)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = []
    for eager in (False, True):
        q = ctx.SimpleQueue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, k, x_dtype, "colshard", q, eager)) for r in range(2)]
        for p in procs:
            p.start()
        out.append(_collect(procs, q))
    for r in range(2):
        assert np.array_equal(out[0][r][0], out[1][r][0]) and np.array_equal(out[0][r][1], out[1][r][1])
        for a, b in zip(out[0][r][2], out[1][r][2]):
            for key in a:
                assert abs(a[key] - b[key]) <= 1e-13 * abs(b[key]), (key, a[key], b[key])


def test_in_library_rccl_step_single_rank():
    """prmf_step_dist (RCCL collectives issued by the library on the stream) with a
    one-rank communicator must reproduce prmf_step bit for bit."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.distributed import DistributedPrmf
    from prmf_amd.solver import PrmfEngine
    case = synth.small_case(seed=17, m=600, n=1100, k=32, P=6, order=70, bf16_x=True)
    X = case["X"]
    res = []
    for native in (False, True):
        eng = PrmfEngine(X.shape[0], X.shape[1], 32, x_dtype="bf16")
        eng.set_x(X)
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(32) % 6)
        U = eng.device_tensor(case["U0"], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        if native:
            runner = DistributedPrmf(eng, native=True)
            assert runner.native
            objs, gd = runner.step(U, V, 4, case["gamma"], case["delta"])
        else:
            objs, gd = eng.step(U, V, 4, case["gamma"], case["delta"])
        res.append((U.cpu().numpy(), V.cpu().numpy(), objs))
        eng.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]


def _native_worker(rank, world, port, k, mode, mock, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PRMF_RCCL_LIB"] = mock
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prmf_amd import PathwayLibrary
        from prmf_amd.distributed import DistributedPrmf
        from prmf_amd.solver import PrmfEngine
        from prmf_amd.synthetic import shard_rows
        case = synth.small_case(seed=13, m=700, n=1500, k=k, P=6, order=80, bf16_x=True)
        X = case["X"]
        lo, hi = shard_rows(X.shape[0], world, rank)
        eng = PrmfEngine(hi - lo, X.shape[1], k, x_dtype="bf16", device=0, nranks=world, rank=rank, dist_mode=mode)
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(k) % 6)
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        runner = DistributedPrmf(eng, native=True)
        assert runner.native                                   # the in-library step, collectives through the mock
        info = eng.comm_info()                                 # what bench.py reports for N > 1
        assert info["in_library"] and info["comm_nranks"] == world and info["comm_rank"] == rank and info["mode"] == mode
        objs, gd = runner.step(U, V, 6, case["gamma"], case["delta"])
        q.put((rank, U.cpu().numpy(), V.cpu().numpy(), objs))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,mode", [(64, "colshard"), (64, "replicated"), (13, "replicated"), (16, "colshard")])
def test_in_library_step_two_ranks_through_mock_rccl(k, mode, tmp_path):
    """prmf_step_dist with TWO ranks: the collectives go through tests/native/mock_rccl.cpp (shared memory between
    the two processes, same entry points and call order as RCCL), so the library's own sequencing -- buffers, counts,
    the auxiliary stream of the Gram all-reduce, the in-place all-reduce of Q -- is checked against the oracle."""
    import subprocess
    import torch.multiprocessing as mp
    from oracle import prmf_oracle as orc
    mock = str(tmp_path / "libmock_rccl.so")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "mock_rccl.cpp"), "-o", mock,
                         "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, port, k, mode, mock, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    U = np.concatenate([res[0][0], res[1][0]])
    case = synth.small_case(seed=13, m=700, n=1500, k=k, P=6, order=80, bf16_x=True)
    pws = [orc.pathway_from_edges(1500, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], [pws[i % 6] for i in range(k)],
                                             n_steps=6, gamma=case["gamma"], delta=case["delta"], trace=tr)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(U, Ur) < 1e-4 and rel(res[0][1], Vr) < 1e-4
    for a, b in zip(res[0][2], tr):
        for key in ("recon", "manifold", "ignore", "fro", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key



def _share_worker(rank, world, port, workload, mode, mock, steps, outdir, q):
    """One rank of a `world`-rank run of BASELINE.json's `workload` (rows sharded as bench.py --gpus N shards them),
    in-library step through the RCCL test double; world == 1: the single-rank prmf_step on the whole matrix."""
    import math
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PRMF_RCCL_LIB"] = mock
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prmf_amd import PathwayLibrary, synthetic as syn
        from prmf_amd.distributed import DistributedPrmf
        from prmf_amd.solver import PrmfEngine
        m, n, k, P, x_dtype = syn.CONFIGS[workload]
        dev = "cuda:0"
        lo, hi = syn.shard_rows(m, world, rank)
        Xd = syn.x_rows_torch(torch, dev, 41, n, k, lo, hi, torch.bfloat16 if x_dtype == "bf16" else torch.float32)
        eng = PrmfEngine(hi - lo, n, k, x_dtype=x_dtype, device=0, nranks=world, rank=rank,
                         dist_mode=mode if world > 1 else None)
        nsq = torch.tensor([eng.local_norm_sq(Xd)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(nsq)
        eng.set_x(Xd, norm_x_sq=float(nsq.item()))
        del Xd
        eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 41)))
        eng.set_active([kk % P for kk in range(k)])
        normx = math.sqrt(float(nsq.item()))
        U = syn.init_factor_torch(torch, dev, 42, m, k, lo, hi)
        V = syn.init_factor_torch(torch, dev, 43, n, k)
        if world > 1:
            runner = DistributedPrmf(eng, native=True)
            assert runner.native and eng.comm_info()["comm_nranks"] == world
            objs, _ = runner.step(U, V, steps, normx / k, 10.0 / normx)
        else:
            objs, _ = eng.step(U, V, steps, normx / k, 10.0 / normx)
        tag = "w%d_r%d" % (world, rank)
        np.save(os.path.join(outdir, "U_%s.npy" % tag), U.cpu().numpy())
        np.save(os.path.join(outdir, "V_%s.npy" % tag), V.cpu().numpy())
        q.put((rank, objs))
        eng.close()
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("workload,mode", [("c3", "colshard"), ("c3", "replicated"), ("c5shard", "colshard")])
def test_in_library_step_two_ranks_at_real_share_shapes(workload, mode, tmp_path):
    """prmf_step_dist with two ranks (RCCL test double) at the shapes a multi-GPU run actually hands a rank: C3 on two
    ranks (25000 x 20000 per rank, k = 64, P = 300: 4 x 64 KiB-row sweeps, two-slice plans, 5 MB exchanges) and a
    25000 x 25000, k = 128, P = 1000 problem on two ranks (12500 x 25000 per rank: the column-split epilogues, 12.8 MB
    exchanges) -- against the single-rank prmf_step on the whole matrix.  The two decompositions sum in different
    orders (float32), so the bar is 1e-5 on U, V and every objective component after 3 iterations; the replicated
    state of the two ranks must agree bit for bit."""
    import subprocess
    import torch.multiprocessing as mp
    from measure import rel, relmax, report
    mock = str(tmp_path / "libmock_rccl.so")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "mock_rccl.cpp"), "-o", mock,
                         "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    ctx = mp.get_context("spawn")
    steps = 3
    out = {}
    for world in (1, 2):
        q = ctx.SimpleQueue()
        port = _free_port()
        procs = [ctx.Process(target=_share_worker, args=(r, world, port, workload, mode, mock, steps, str(tmp_path), q))
                 for r in range(world)]
        for p in procs:
            p.start()
        out[world] = _collect(procs, q, deadline_s=600)
    ld = lambda name: np.load(str(tmp_path / name))
    V0, V1 = ld("V_w2_r0.npy"), ld("V_w2_r1.npy")
    assert np.array_equal(V0, V1) and out[2][0][0] == out[2][1][0]
    U2 = np.concatenate([ld("U_w2_r0.npy"), ld("U_w2_r1.npy")])
    eu, ev = rel(U2, ld("U_w1_r0.npy")), rel(V0, ld("V_w1_r0.npy"))
    eo = {key: relmax([o[key] for o in out[2][0][0]], [o[key] for o in out[1][0][0]])
          for key in ("recon", "manifold", "ignore", "fro", "obj")}
    report("two_ranks_real_share[%s-%s]" % (workload, mode), steps=steps, rel_U=eu, rel_V=ev, **{"rel_" + k_: v for k_, v in eo.items()})
    assert eu < 1e-5 and ev < 1e-5
    for key, v in eo.items():
        assert v < 1e-5, (key, v)


def _empty_rank_worker(rank, world, port, mode, mock, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mock:
        os.environ["PRMF_RCCL_LIB"] = mock
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prmf_amd import PathwayLibrary
        from prmf_amd.distributed import DistributedPrmf
        from prmf_amd.solver import PrmfEngine
        from prmf_amd.synthetic import shard_rows
        case = synth.small_case(seed=19, m=1, n=90, k=4, P=3, order=12, zero_genes=2, isolated=1)
        X = case["X"]
        lo, hi = shard_rows(X.shape[0], world, rank)           # rank 0: the one sample; rank 1: no rows at all
        eng = PrmfEngine(hi - lo, X.shape[1], 4, x_dtype="f32", device=0, nranks=world, rank=rank, dist_mode=mode)
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(4) % 3)
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        objs, gd = DistributedPrmf(eng, native=True if mock else None).step(U, V, 5, case["gamma"], case["delta"])
        q.put((rank, U.cpu().numpy(), V.cpu().numpy(), objs))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,native", [("colshard", True), ("replicated", True), ("colshard", False), ("replicated", False)])
def test_rank_without_samples(mode, native, tmp_path):
    """One sample on two ranks: rank 1 holds no rows of X (no sweeps, zero contributions to both exchanges) but still
    updates its share of V and keeps the same records as rank 0."""
    import subprocess
    import torch.multiprocessing as mp
    from oracle import prmf_oracle as orc
    mock = None
    if native:
        mock = str(tmp_path / "libmock_rccl.so")
        cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                             os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "mock_rccl.cpp"), "-o", mock,
                             "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-2000:]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, 2, port, mode, mock, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q)
    assert res[1][0].shape[0] == 0 and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    case = synth.small_case(seed=19, m=1, n=90, k=4, P=3, order=12, zero_genes=2, isolated=1)
    pws = [orc.pathway_from_edges(90, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], [pws[i % 3] for i in range(4)], n_steps=5,
                                             gamma=case["gamma"], delta=case["delta"], trace=tr)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(res[0][0], Ur) < 1e-4 and rel(res[0][1], Vr) < 1e-4
    for a, b in zip(res[0][2], tr):
        for key in ("recon", "manifold", "ignore", "fro", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key


def _tradeoff_worker(rank, world, port, mode, mock, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mock:
        os.environ["PRMF_RCCL_LIB"] = mock
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from prmf_amd import PathwayLibrary
        from prmf_amd.distributed import DistributedPrmf
        from prmf_amd.solver import PrmfEngine
        from prmf_amd.synthetic import shard_rows
        case = synth.small_case(seed=23, m=301, n=800, k=8, P=4, order=60)
        X = case["X"]
        lo, hi = shard_rows(X.shape[0], world, rank)
        eng = PrmfEngine(hi - lo, X.shape[1], 8, x_dtype="f32", device=0, nranks=world, rank=rank, dist_mode=mode)
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(8) % 4)
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        objs, gd = DistributedPrmf(eng, native=True if mock else None).step(U, V, 6, case["gamma"], case["delta"], tradeoff=0.4)
        q.put((rank, U.cpu().numpy(), V.cpu().numpy(), objs, gd, list(eng.last_flags)))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,native", [("colshard", True), ("replicated", True), ("colshard", False)])
def test_two_ranks_tradeoff_feedback(mode, native, tmp_path):
    """gamma = delta = (1 - t) recon / (t manifold) (prmf_runner.py:542-548) on two ranks: the per-iteration objective,
    with the in-library step every record from the direct residual (the ranks' sums all-reduced each iteration), host-driven
    from the Gram identity -- against the oracle's tradeoff loop."""
    import subprocess
    import torch.multiprocessing as mp
    from oracle import prmf_oracle as orc
    mock = None
    if native:
        mock = str(tmp_path / "libmock_rccl.so")
        cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                             os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "mock_rccl.cpp"), "-o", mock,
                             "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-2000:]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_tradeoff_worker, args=(r, 2, port, mode, mock, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2] and res[0][3] == res[1][3]
    assert all(bool(f & 1) == native for f in res[0][4]), res[0][4]       # direct residual exactly when the library can sum the ranks
    case = synth.small_case(seed=23, m=301, n=800, k=8, P=4, order=60)
    pws = [orc.pathway_from_edges(800, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od, g_ref, d_ref = orc.nmf_manifold_vec_update_tradeoff(case["X"], case["U0"], case["V0"], [pws[i % 4] for i in range(8)],
                                                                   n_steps=6, tradeoff=0.4, gamma=case["gamma"], delta=case["delta"], trace=tr)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    U = np.concatenate([res[0][0], res[1][0]])
    assert rel(U, Ur) < 1e-4 and rel(res[0][1], Vr) < 1e-4
    for a, b in zip(res[0][2], tr):
        for key in ("recon", "manifold", "ignore", "fro", "gamma", "delta", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key
    assert abs(res[0][3][0] - g_ref) <= 1e-4 * abs(g_ref)
