"""The 8-way partition of BASELINE.json configs[3-4], executed.

configs[3] = the 50000 x 20000, k = 64 problem with its latent columns sharded 8 ways (6250-row shares, kc = 8);
configs[4] = 200000 x 25000, k = 128 on 8 ranks (25000-row shares, kc = 16).  The test box has ONE GPU and allows few
processes on it, so the ranks are THREADS of one worker process (prmf_amd.ranks.LocalRanks: the single-process
multi-device form) that share cuda:0, and the collectives of the in-library step (prmf_step_dist) go through the RCCL
test double (tests/native/mock_rccl.cpp: same entry points, same call order, shared memory instead of xGMI) -- real RCCL
refuses two ranks on one device.  What runs is everything but the wire: every kernel of every rank at the real share
shapes, the exchange buffers, counts and layouts of 4 and 8 rank blocks, ranks that own no latent column, shard sizes
that differ by one, the CLI and bench.py on 8 ranks, and what happens when a collective fails.  There is no reference
counterpart (script/prmf_runner.py:419-449 is one process); the bar is the single-rank prmf_step on the whole matrix and
the reference goldens the single-rank solver already meets.
"""
import json
import math
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import synth
from measure import rel, relmax, report

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("recon", "manifold", "ignore", "fro", "obj")


@pytest.fixture(scope="module")
def mock(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("mock_rccl") / "libmock_rccl.so")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(ROOT, "tests", "native", "mock_rccl.cpp"), "-o", path,
                         "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    return path


def _in_worker(target, args, env, deadline_s=900):
    """Run `target(*args, q)` in ONE fresh process (the library binds its RCCL -- here the double -- once per process, and
    the pytest process may already have bound the real one) with `env` added to its environment; -> what it put in q."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        p = ctx.Process(target=target, args=tuple(args) + (q,))
        p.start()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    t0 = time.time()
    res = None
    while res is None:
        if not q.empty():
            res = q.get()
            break
        if p.exitcode not in (None, 0) or time.time() - t0 > deadline_s:
            if p.is_alive():
                p.kill()
            raise AssertionError("worker failed (exit code %s) or timed out after %.0f s" % (p.exitcode, time.time() - t0))
        time.sleep(0.05)
    p.join(120)
    if p.is_alive():
        p.kill()
    assert p.exitcode == 0
    return res


# ---------------------------------------------------------------------------------------------------------------------
# (a) prmf_step_dist with 4 and 8 ranks at the real share shapes
# ---------------------------------------------------------------------------------------------------------------------
def _share_worker(workload, variants, steps, q):
    import torch
    from prmf_amd import PathwayLibrary, synthetic as syn
    from prmf_amd.distributed import DistributedPrmf
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import PrmfEngine
    m, n, k, P, x_dtype = syn.CONFIGS[workload]
    tdt = torch.bfloat16 if x_dtype == "bf16" else torch.float32
    lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 41))
    active = [kk % P for kk in range(k)]

    def solve(view):
        world, rank = (view.world, view.rank) if view is not None else (1, 0)
        lo, hi = syn.shard_rows(m, world, rank)
        Xd = syn.x_rows_torch(torch, "cuda:0", 41, n, k, lo, hi, tdt)
        # the one-rank reference at k > 64 takes EVERY record's recon from the direct residual (prmf_runner.py:337 itself): what
        # the comparison then shows is the multi-rank run's own identity error, not two identity errors of different chains
        eng = PrmfEngine(hi - lo, n, k, x_dtype=x_dtype, device=0, nranks=world, rank=rank, dist_mode=view.mode if view is not None else None,
                         direct_residual=(view is None and k > 64))
        local = eng.local_norm_sq(Xd)
        nsq = view.sum_float(local) if view is not None else local
        eng.set_x(Xd, norm_x_sq=nsq)
        del Xd
        eng.set_library(lib)
        eng.set_active(active)
        normx = math.sqrt(nsq)
        U = syn.init_factor_torch(torch, "cuda:0", 42, m, k, lo, hi)
        V = syn.init_factor_torch(torch, "cuda:0", 43, n, k)
        if view is not None:
            runner = DistributedPrmf(eng, group=view, native=True)
            info = eng.comm_info()
            assert runner.native and info["in_library"] and info["comm_nranks"] == world and info["comm_rank"] == rank
            assert eng.sizes.kc == -(-k // (1 if view.mode == "replicated" else world))
            objs, _ = runner.step(U, V, steps, normx / k, 10.0 / normx)
        else:
            objs, _ = eng.step(U, V, steps, normx / k, 10.0 / normx)
        out = (U.cpu().numpy(), V.cpu().numpy(), objs, list(eng.last_flags))
        eng.close()
        return out

    U1, V1, objs1, _ = solve(None)                      # the whole matrix on one rank: prmf_step
    torch.cuda.empty_cache()
    result = {}
    for world, mode in variants:
        def body(view, mode=mode):
            view.mode = mode
            return solve(view)
        t0 = time.time()
        res = LocalRanks(world).run(body)
        torch.cuda.empty_cache()
        Vs = [r[1] for r in res]
        same = all(np.array_equal(Vs[0], v) for v in Vs[1:]) and all(r[2] == res[0][2] and r[3] == res[0][3] for r in res[1:])
        Un = np.concatenate([r[0] for r in res])
        result["%d-%s" % (world, mode)] = {
            "replicated_state_identical": bool(same), "rel_U": rel(Un, U1), "rel_V": rel(Vs[0], V1),
            "rel_obj": {key: relmax([o[key] for o in res[0][2]], [o[key] for o in objs1]) for key in KEYS},
            "rows": [int(r[0].shape[0]) for r in res], "v_min": float(Vs[0].min()), "seconds": time.time() - t0}
    q.put(result)


@pytest.mark.parametrize("workload,variants,steps", [
    ("c3", [(8, "colshard"), (8, "replicated"), (4, "colshard")], 3),      # BASELINE configs[3]: 6250 x 20000 per rank, kc = 8
    ("c5", [(8, "colshard"), (8, "replicated"), (4, "colshard")], 3),      # BASELINE configs[4]: 25000 x 25000 per rank, kc = 16
    ("c3", [(8, "colshard")], 30),                                         # three rounds' worth: the sharded run does not drift away
])
def test_step_dist_four_and_eight_ranks_at_real_share_shapes(workload, variants, steps, mock):
    """prmf_step_dist on 8 (and 4) ranks, both V-update modes, 3 iterations (and 30 on 8 ranks), against the single-rank
    prmf_step on the whole matrix (the same kernels summing in another order: float32, so 1e-5 as for two ranks); the
    replicated state -- V, every objective record, the flags -- must agree bit for bit across the ranks."""
    out = _in_worker(_share_worker, (workload, variants, steps), {"PRMF_RCCL_LIB": mock}, deadline_s=1500)
    from prmf_amd import synthetic as syn
    m = syn.CONFIGS[workload][0]
    for world, mode in variants:
        r = out["%d-%s" % (world, mode)]
        report("step_dist_real_shares[%s-%d-%s]" % (workload, world, mode), steps=steps, rel_U=r["rel_U"], rel_V=r["rel_V"],
               seconds=r["seconds"], **{"rel_" + k_: v for k_, v in r["rel_obj"].items()})
        assert r["rows"] == [syn.shard_rows(m, world, q)[1] - syn.shard_rows(m, world, q)[0] for q in range(world)]
        assert r["replicated_state_identical"]
        assert r["rel_U"] < 1e-5 and r["rel_V"] < 1e-5 and r["v_min"] >= np.finfo(np.float32).eps
        # k > 64: the identity-based recon of the sharded run carries ~1e-7 ||X||^2 of float32 accumulation error over its
        # 25000- / 50000-sample chains (1e-5 / 2e-5 on recon at this fit); the reference's records are exact
        tol = 1e-5 if syn.CONFIGS[workload][2] <= 64 else 4e-5
        for key, v in r["rel_obj"].items():
            assert v < (tol if key in ("recon", "obj") else 1e-5), (world, mode, key, v)


def _small_worker(world, k, x_dtype, mode, m, q, tradeoff=None, steps=6):
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.distributed import DistributedPrmf
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import PrmfEngine
    from prmf_amd.synthetic import shard_rows
    case = synth.small_case(seed=13, m=m, n=1500, k=k, P=6, order=80, bf16_x=(x_dtype == "bf16"))
    X = case["X"]

    def body(view):
        lo, hi = shard_rows(X.shape[0], view.world, view.rank)
        eng = PrmfEngine(hi - lo, X.shape[1], k, x_dtype=x_dtype, device=0, nranks=view.world, rank=view.rank, dist_mode=mode)
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(k) % 6)
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        objs, gd = DistributedPrmf(eng, group=view, native=True).step(U, V, steps, case["gamma"], case["delta"], tradeoff)
        out = (U.cpu().numpy(), V.cpu().numpy(), objs, eng.sizes.kc, (gd, list(eng.last_flags)))
        eng.close()
        return out

    q.put([LocalRanks(world, timeout_s=300).run(body) for _ in range(2)])      # twice: the sharded step is reproducible bit for bit


@pytest.mark.parametrize("world,k,x_dtype,mode,m", [(8, 13, "bf16", "colshard", 700), (8, 13, "bf16", "replicated", 700), (8, 100, "f32", "colshard", 700),
                                                    (8, 6, "f32", "colshard", 5), (6, 64, "bf16", "colshard", 333), (8, 128, "bf16", "replicated", 700),
                                                    (5, 96, "f32", "colshard", 333), (5, 96, "bf16", "colshard", 4)])
def test_step_dist_eight_ranks_small_shapes_vs_oracle(mock, world, k, x_dtype, mode, m):
    """Ragged everything on 8 (and 6) ranks through the in-library step: k = 13 (kc = 2: one rank with one column, one with
    none), k = 100 and 128 (the column-split epilogues, kc = 13 / 16), 5 samples on 8 ranks (three ranks without rows),
    87- and 88-row shards; k = 96 on FIVE ranks (kc = 20: the five blocks together are 100 columns, wider than the 96-column
    tile -- the assembling kernel's staged copy did not fit and the ranks disagreed until round 6's fuzzer found it; once with a
    rank that holds no sample) -- against the oracle, and run twice: bit-identical."""
    from oracle import prmf_oracle as orc
    runs = _in_worker(_small_worker, (world, k, x_dtype, mode, m), {"PRMF_RCCL_LIB": mock, "MOCK_RCCL_TIMEOUT_S": "60"}, deadline_s=600)
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    res = runs[0]
    for other in res[1:]:
        assert np.array_equal(other[1], res[0][1]) and other[2] == res[0][2]
    assert res[0][3] == (k if mode == "replicated" else -(-k // world))
    U = np.concatenate([r[0] for r in res])
    case = synth.small_case(seed=13, m=m, n=1500, k=k, P=6, order=80, bf16_x=(x_dtype == "bf16"))
    pws = [orc.pathway_from_edges(1500, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], [pws[i % 6] for i in range(k)], n_steps=6,
                                             gamma=case["gamma"], delta=case["delta"], trace=tr)
    eu, ev = rel(U, Ur), rel(res[0][1], Vr)
    eo = max(relmax([o[key] for o in res[0][2]], [t[key] for t in tr]) for key in KEYS)
    report("step_dist_small_shapes[%d-%d-%s-%s-m%d]" % (world, k, x_dtype, mode, m), rel_U=eu, rel_V=ev, rel_obj=eo)
    assert eu < 1e-4 and ev < 1e-4 and eo < 1e-4


def _small_tradeoff_worker(world, k, mode, m, steps, q):
    _small_worker(world, k, "f32", mode, m, q, tradeoff=0.4, steps=steps)


# (five samples: gamma runs away in the reference itself -- 1.8e44 at step 6, beyond float32 -- so three steps there)
@pytest.mark.parametrize("world,k,mode,m,steps", [(8, 8, "colshard", 301, 6), (8, 8, "replicated", 5, 3)])
def test_tradeoff_feedback_on_eight_ranks(mock, world, k, mode, m, steps):
    """gamma = delta = (1 - t) recon / (t manifold) (prmf_runner.py:542-548) through the in-library step on 8 ranks: every
    record from the direct residual -- each rank's sum (X_g - U_g V^T)^2 all-reduced every iteration, ranks without rows
    (5 samples on 8 ranks) contributing zero -- against the oracle's tradeoff loop; run twice: bit-identical.

    What the 1e-4 on the final gamma means here (round 6, profiles/r06_tradeoff_digits.txt, r06_first_step_errors.txt): in the
    301-sample case gamma runs from 12 to 2e33 in six steps and the recursion multiplies an error of the state by ~3 per step
    (the float64 oracle itself turns a 1e-7 relative perturbation of U0, V0 into up to 7e-6 on the final gamma).  Through round
    5 the library landed at 9.3e-5 -- one rank and eight alike, so nothing was lost in the exchanges: the digits went where the
    manifold term is normalised, because the sweeps multiply with factors packed to 16-17 mantissa bits and over 301 samples
    that rounding does not average out (every gene of a latent column gets the same relative shift, the column norm moves with
    it).  Since round 6 a reduction this short is multiplied exactly (csrc/gemm.hip: exact_product_kernel): 3.0e-5 on eight
    ranks, 1.6e-5 on one."""
    from oracle import prmf_oracle as orc
    runs = _in_worker(_small_tradeoff_worker, (world, k, mode, m, steps), {"PRMF_RCCL_LIB": mock, "MOCK_RCCL_TIMEOUT_S": "60"}, deadline_s=600)
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[4] == b[4]
    res = runs[0]
    for other in res[1:]:
        assert np.array_equal(other[1], res[0][1]) and other[2] == res[0][2] and other[4] == res[0][4]
    gd, flags = res[0][4]
    assert all(f & 1 for f in flags), flags                       # PRMF_RECON_DIRECT on every record
    U = np.concatenate([r[0] for r in res])
    case = synth.small_case(seed=13, m=m, n=1500, k=k, P=6, order=80)
    pws = [orc.pathway_from_edges(1500, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    tr = []
    Ur, Vr, od, g_ref, d_ref = orc.nmf_manifold_vec_update_tradeoff(case["X"], case["U0"], case["V0"], [pws[i % 6] for i in range(k)], n_steps=steps,
                                                                   tradeoff=0.4, gamma=case["gamma"], delta=case["delta"], trace=tr)
    eu, ev = rel(U, Ur), rel(res[0][1], Vr)
    eo = max(relmax([o[key] for o in res[0][2]], [t[key] for t in tr]) for key in KEYS + ("gamma", "delta"))
    report("tradeoff_eight_ranks[%s-m%d]" % (mode, m), rel_U=eu, rel_V=ev, rel_obj=eo, rel_gamma=abs(gd[0] - g_ref) / abs(g_ref))
    assert eu < 1e-4 and ev < 1e-4 and eo < 1e-4 and abs(gd[0] - g_ref) <= 1e-4 * abs(g_ref)


# ---------------------------------------------------------------------------------------------------------------------
# (b) nmf_pathway / the CLI on 8 ranks: k = 6 on 8 ranks (ranks 6 and 7 own no latent column), 100 samples (13- and
#     12-row shards)
# ---------------------------------------------------------------------------------------------------------------------
def _pathway_worker(casedir, world, mode, q):
    import networkx as nx
    from prmf_amd import runner
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import nmf_pathway
    c = synth.cli_case(casedir, variant=1)
    graphs = [runner.relabel_nodes(nx.read_graphml(fp).to_undirected(), "name") for fp in c["manifolds"]]
    np.random.seed(1)                                    # rank 0's draws are the solve's; no other rank draws

    def body(view):
        rounds, trace = [], []
        U, V, od = nmf_pathway(c["X"], graphs, nodelist=c["nodelist_ids"], gamma=1.0, k_latent=6, round_log=rounds, trace=trace,
                               dist_mode=mode, group=view)
        l2p = od["latent_to_pathway_data"]
        return U, V, [t["obj"] for t in trace], rounds, [int(l2p[k][0][0]) for k in sorted(l2p)]

    q.put(LocalRanks(world).run(body))


@pytest.mark.parametrize("mode", ["colshard", "replicated"])
def test_nmf_pathway_on_eight_ranks_vs_reference(golden_dir, tmp_path, mock, mode):
    g = np.load(os.path.join(golden_dir, "g6_api.npz"))
    casedir = str(tmp_path / "case")
    os.makedirs(casedir)
    res = _in_worker(_pathway_worker, (casedir, 8, mode), {"PRMF_RCCL_LIB": mock}, deadline_s=900)
    U0, V0, tr0, rounds0, map0 = res[0]
    for U, V, tr, rounds, mp_ in res[1:]:
        assert np.array_equal(U0, U) and np.array_equal(V0, V) and tr0 == tr and rounds0 == rounds and map0 == mp_
    e_tr, eu, ev = relmax(tr0, g["trace"][:len(tr0)]), rel(U0, g["U"]), rel(V0, g["V"])
    report("nmf_pathway_eight_ranks[%s]" % mode, iterations=len(tr0), rel_trace=e_tr, rel_U=eu, rel_V=ev)
    assert U0.shape == g["U"].shape == (100, 6)
    assert np.array_equal(np.array(rounds0), g["rounds"]) and len(tr0) == len(g["trace"]) and map0 == list(g["final_map"])
    assert e_tr < 1e-4 and eu < 1e-4 and ev < 1e-4


def _read_obj(fp):
    vals, mapping = {}, []
    for line in open(fp):
        if " = " in line:
            a, b = line.rstrip().split(" = ")
            vals[a] = float(b)
        elif "->" in line:
            a, b = line.rstrip().split("->")
            mapping.append([int(a), os.path.basename(b.strip())])
    return vals, mapping


@pytest.mark.parametrize("norm", ["nonorm", "norm"])
def test_cli_eight_ranks_vs_reference(golden_dir, tmp_path, mock, norm):
    """`prmf_runner.py --n-gpus 8 --launcher threads` with the CLI's default -k 6: eight ranks in one process, rank 0
    writes the reference's files."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "g6_%s.npz" % norm))
    c = synth.cli_case(str(tmp_path), variant=1)
    out = str(tmp_path / "out")
    os.makedirs(out)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py"), "--n-gpus", "8", "--launcher", "threads",
                        "--data", c["data"], "--manifolds"] + c["manifolds"] +
                       ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--delimiter", "\t", "--seed", "1"] +
                       (["--no-normalize"] if norm == "nonorm" else []),
                       cwd=str(tmp_path), capture_output=True, text=True, env=dict(os.environ, PRMF_RCCL_LIB=mock), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if len(l.split()) == 2 and l.split()[0].isdigit()]
    trace = np.array([float(l.split()[1]) for l in lines])
    assert r.stdout.count("norm(X) = ") == 1                     # rank 0 alone prints
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    ref = json.loads(str(g["obj_txt"]))
    assert list(vals) == list(ref)
    U = pd.read_csv(os.path.join(out, "U.csv"))
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0)
    eu, ev, et = rel(U.to_numpy(), g["U"]), rel(V.to_numpy(), g["V"]), relmax(trace, g["trace"][:len(trace)])
    report("cli_eight_ranks[%s]" % norm, iterations=len(trace), rel_trace=et, rel_U=eu, rel_V=ev)
    assert len(trace) == len(g["trace"]) and mapping == json.loads(str(g["mapping"]))
    assert et < 1e-4 and eu < 1e-4 and ev < 1e-4
    for key in ref:
        assert abs(vals[key] - ref[key]) <= 1e-4 * abs(ref[key]) + 1e-5, key


# ---------------------------------------------------------------------------------------------------------------------
# (c) bench.py --gpus 8, rehearsed
# ---------------------------------------------------------------------------------------------------------------------
def test_bench_eight_rank_rehearsal(mock):
    """`bench.py --gpus 8` started plainly = 8 thread ranks of the C3 problem (6250 x 20000 per rank) on the one GPU, the
    in-library RCCL step through the double: the line must carry the `distributed` block with the communicator size RCCL
    (here: the double) itself reports."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "c3", "--steps", "4", "--warmup", "2",
                        "--reps", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, PRMF_RCCL_LIB=mock), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = line["distributed"]
    assert line["n_gpus"] == 8 and d["rccl_comm_world_size"] == 8 and d["world_size"] == 8 and d["mode"] == "colshard"
    assert d["collectives"].startswith("in-library") and "threads" in d["ranks_are"] and "rehearsal" in line["config"]
    assert line["value"] > 0 and math.isfinite(line["recon_err"]) and line["steps"] == 4
    assert all(v > 0 for v in d["phase_ms_rank0"].values())
    report("bench_eight_rank_rehearsal", recon_err=line["recon_err"], ms_per_step_shared_gpu=line["ms_per_step"])


def test_bench_under_the_process_launcher(mock):
    """The driver's launch line for N > 1 -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- with
    four process ranks sharing the GPU (gloo for torch.distributed, the in-library step through the double): rank 0's
    one JSON line, ranks = processes, the communicator the double reports, max-over-ranks timing."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, PRMF_RCCL_LIB=mock, PRMF_BENCH_BACKEND="gloo", PRMF_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "c2", "--steps", "6",
                        "--warmup", "2", "--reps", "2"], cwd=ROOT, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    line = json.loads(lines[0])
    d = line["distributed"]
    assert line["n_gpus"] == 4 and d["rccl_comm_world_size"] == 4 and "processes" in d["ranks_are"] and d["collectives"].startswith("in-library")
    assert line["scaling"] == "strong" and line["value"] > 0 and math.isfinite(line["recon_err"]) and "cpu_baseline" not in line
    report("bench_process_launcher_rehearsal", recon_err=line["recon_err"], ms_per_step_shared_gpu=line["ms_per_step"])


# ---------------------------------------------------------------------------------------------------------------------
# (d) a collective fails
# ---------------------------------------------------------------------------------------------------------------------
def _failure_worker(world, mode, scenario, mock, q):
    import ctypes
    import torch
    from prmf_amd import PathwayLibrary, _capi
    from prmf_amd.distributed import DistributedPrmf
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import PrmfEngine
    from prmf_amd.synthetic import shard_rows
    case = synth.small_case(seed=13, m=700, n=1500, k=16, P=6, order=80, bf16_x=True)
    X = case["X"]
    mocklib = ctypes.CDLL(mock)

    def body(view):
        lo, hi = shard_rows(X.shape[0], view.world, view.rank)
        eng = PrmfEngine(hi - lo, X.shape[1], 16, x_dtype="bf16", device=0, nranks=view.world, rank=view.rank, dist_mode=mode)
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(16) % 6)
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        runner = DistributedPrmf(eng, group=view, native=True)
        if scenario == "stream_timeout":
            # a stream that stays busy (stand-in for a collective kernel whose peer never arrives): the bounded wait
            # must come back with an error instead of hanging, and the wait with a generous limit must succeed
            eng.set_comm_timeout(60.0)
            assert eng.lib.prmf_debug_sync_stream(eng.ctx, eng._stream()) == 0
            a = torch.randn((8192, 8192), device="cuda:0")
            eng.set_comm_timeout(0.05)
            t0 = time.time()
            for _ in range(150):
                a = (a @ a).clamp_(-1.0, 1.0)
            rc = eng.lib.prmf_debug_sync_stream(eng.ctx, eng._stream())
            waited = time.time() - t0
            msg = eng.lib.prmf_last_error(eng.ctx).decode()
            torch.cuda.synchronize()
            busy = time.time() - t0
            return rc, msg, waited, busy
        if scenario == "peer_gone" and view.rank == 1:
            return "left", "", 0.0, 0                     # this rank never enters a collective again
        t0 = time.time()
        try:
            runner.step(U, V, 6, case["gamma"], case["delta"])
            first = (0, "")
        except _capi.PrmfError as e:
            first = (e.code, str(e))
        waited = time.time() - t0
        depth = mocklib.mock_rccl_group_depth()            # thread-local in the double: an open group would show here
        try:
            runner.step(U, V, 1, case["gamma"], case["delta"])
            second = 0
        except _capi.PrmfError as e:
            second = e.code
        return first[0], first[1], waited, (depth, second)

    q.put(LocalRanks(world, timeout_s=120).run(body))


@pytest.mark.parametrize("mode,fail", [("colshard", "2:4"), ("replicated", "1:2"), ("colshard", "3:6")])
def test_failed_collective_ends_every_rank_with_an_error(mock, mode, fail):
    """One rank's collective returns an error (MOCK_RCCL_FAIL = rank:call; call 0 is prmf_comm_init's all-reduce of the
    sample count, then 3 (colshard) or 2 (replicated) per iteration -- "2:4" is the k x k all-reduce that opens the RCCL
    group of iteration 2, "1:2" the all-reduce of X^T U inside the group of iteration 1, "3:6" the all-gather of
    iteration 2): that rank closes the group, aborts its communicator and returns
    PRMF_ERR_HIP; the abort reaches the peers waiting for it, which return PRMF_ERR_HIP too.  Nobody hangs, no group is
    left open, and the context refuses further multi-rank steps (PRMF_ERR_STATE)."""
    t0 = time.time()
    res = _in_worker(_failure_worker, (4, mode, "fail", mock), {"PRMF_RCCL_LIB": mock, "MOCK_RCCL_FAIL": fail, "MOCK_RCCL_TIMEOUT_S": "60"},
                     deadline_s=300)
    bad = int(fail.split(":")[0])
    for rank, (code, msg, waited, (depth, second)) in enumerate(res):
        assert code == -2, (rank, code, msg)                                     # PRMF_ERR_HIP
        assert ("injected failure" in msg) == (rank == bad), (rank, msg)
        assert "aborted" in msg and depth == 0 and second == -3 and waited < 30, (rank, msg, depth, second, waited)
    report("failed_collective[%s-%s]" % (mode, fail), seconds=time.time() - t0, max_wait_s=max(r[2] for r in res))


def test_peer_that_never_arrives_does_not_hang_the_others(mock):
    """A rank that is gone (it simply never calls the step) leaves its peers inside the first collective: the double's
    arrival timeout (3 s here; with RCCL: the library's stream timeout, next test) turns that into PRMF_ERR_HIP."""
    res = _in_worker(_failure_worker, (3, "colshard", "peer_gone", mock), {"PRMF_RCCL_LIB": mock, "MOCK_RCCL_TIMEOUT_S": "3"}, deadline_s=300)
    assert res[1][0] == "left"
    for rank in (0, 2):
        code, msg, waited, (depth, second) = res[rank]
        assert code == -2 and "aborted" in msg and depth == 0 and second == -3 and waited < 60, res[rank]


def test_bounded_stream_wait(mock):
    """prmf_set_comm_timeout: the library's wait for a stream that carries collectives returns PRMF_ERR_HIP after the
    limit (the communicator is aborted) instead of blocking in hipStreamSynchronize for as long as the stream is busy."""
    res = _in_worker(_failure_worker, (2, "colshard", "stream_timeout", mock), {"PRMF_RCCL_LIB": mock}, deadline_s=300)
    for rc, msg, waited, busy in res:
        assert rc == -2 and "timed out" in msg and "aborted" in msg, (rc, msg)
        assert waited < busy and waited < 0.5 * busy + 1.0, (waited, busy)
    report("bounded_stream_wait", returned_after_s=max(r[2] for r in res), stream_busy_s=max(r[3] for r in res))


def test_cli_job_dies_nonzero_when_a_rank_fails(tmp_path, mock):
    """The process launcher: `prmf_runner.py --n-gpus 2` with one rank's 8th collective failing -- that rank exits
    non-zero, its peer gets the abort and exits non-zero, torch.distributed.run ends the job: a non-zero exit code within
    the deadline, no hang, no output files."""
    c = synth.cli_case(str(tmp_path), variant=1)
    out = str(tmp_path / "out")
    os.makedirs(out)
    env = dict(os.environ, PRMF_DIST_BACKEND="gloo", PRMF_SHARE_GPU="1", PRMF_RCCL_LIB=mock, MOCK_RCCL_FAIL="1:7", MOCK_RCCL_TIMEOUT_S="30")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py"), "--n-gpus", "2", "--data", c["data"],
                        "--manifolds"] + c["manifolds"] + ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out,
                                                           "--delimiter", "\t", "--seed", "1", "--no-normalize"],
                       cwd=str(tmp_path), capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "injected failure" in r.stderr and not os.path.exists(os.path.join(out, "U.csv"))
    report("cli_job_dies_nonzero", seconds=time.time() - t0, returncode=r.returncode)


# ---------------------------------------------------------------------------------------------------------------------
# the direct-residual decision must not depend on the rank's own row count
# ---------------------------------------------------------------------------------------------------------------------
def _threshold_worker(eps, steps, q):
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.distributed import DistributedPrmf
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import PrmfEngine
    from prmf_amd.synthetic import shard_rows
    from test_gpu_recon import direct_fp64, lowrank_case
    m, n, k = 3, 400, 8
    case = lowrank_case(5, m, n, k, eps)
    U0, V0 = np.abs(case["U0"]) + 1e-3, np.abs(case["V0"]) + 1e-3

    def body(view):
        lo, hi = shard_rows(m, view.world, view.rank)          # 2 rows and 1 row
        eng = PrmfEngine(hi - lo, n, k, x_dtype="f32", device=0, nranks=2, rank=view.rank, dist_mode=view.mode)
        Xd = eng.device_tensor(case["X"][lo:hi], torch.float32)
        eng.set_x(Xd, norm_x_sq=float(np.sum(case["X"] * case["X"])))
        eng.set_library(PathwayLibrary.from_edge_lists(n, case["pathways"]))
        eng.set_active(np.arange(k) % len(case["pathways"]))
        U = eng.device_tensor(U0[lo:hi], torch.float32)
        V = eng.device_tensor(V0, torch.float32)
        objs, _ = DistributedPrmf(eng, group=view, native=True).step(U, V, steps, 1e-3, 1e-3)
        return objs, list(eng.last_flags), direct_fp64(torch, Xd, U, V)

    out = {}
    for mode in ("colshard", "replicated"):
        def run(view, mode=mode):
            view.mode = mode
            return body(view)
        out[mode] = LocalRanks(2, timeout_s=120).run(run)
    q.put((out, float(np.sum(case["X"] * case["X"]))))


@pytest.mark.parametrize("eps,steps,expect_direct", [(1.2, 1, False), (0.5, 3, True)])
def test_direct_residual_decision_is_rank_invariant(mock, eps, steps, expect_direct):
    """Three samples on two ranks (2 rows + 1 row), k = 8.  The threshold below which the last record takes the direct
    residual contains 0.13 / sqrt(min(m, n) k): evaluated with each rank's OWN row count it was 0.033 ||X||^2 on rank 0
    and 0.046 ||X||^2 on rank 1, and a fit of 0.041 ||X||^2 (eps = 1.2, one step) made rank 1 enter an all-reduce that
    rank 0 never joined.  With the global m (0.027 ||X||^2) every rank decides alike: nobody at 0.041, everybody at
    0.006 (eps = 0.5, three steps) -- where the record must then equal the float64 residual of the ranks' own factors."""
    out, nsq = _in_worker(_threshold_worker, (eps, steps), {"PRMF_RCCL_LIB": mock, "MOCK_RCCL_TIMEOUT_S": "20"}, deadline_s=300)
    for mode, ((objs0, flags0, own0), (objs1, flags1, own1)) in out.items():
        assert objs0 == objs1 and flags0 == flags1, mode
        frac = objs0[-1]["recon"] ** 2 / nsq
        assert bool(flags0[-1] & 1) == expect_direct, (mode, flags0, frac)
        assert (0.033 < frac < 0.046) if not expect_direct else frac < 0.027, (mode, frac)
        if expect_direct:
            own = math.sqrt(own0 + own1)
            assert abs(objs0[-1]["recon"] - own) < 1e-5 * own
        report("rank_invariant_direct_decision[%s-eps%g]" % (mode, eps), recon_sq_over_normx_sq=frac, flags=flags0[-1])


# ---------------------------------------------------------------------------------------------------------------------
# host-driven collectives between thread ranks (no RCCL at all): the group's own device-tensor exchanges on the GPU
# ---------------------------------------------------------------------------------------------------------------------
def _host_driven_case(k, tradeoff):
    # with the feedback on: the problem of test_gpu_distributed.py::test_two_ranks_tradeoff_feedback (gamma grows 1e4-fold in
    # five steps there; on a larger one the host-driven path's identity-based recon leaves gamma 1.007e-4 off after five)
    if tradeoff is not None:
        return synth.small_case(seed=23, m=301, n=800, k=k, P=4, order=60)
    return synth.small_case(seed=13, m=301, n=1500, k=k, P=6, order=80)


def _host_driven_worker(world, k, mode, tradeoff, q):
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd.distributed import DistributedPrmf
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import PrmfEngine
    from prmf_amd.synthetic import shard_rows
    case = _host_driven_case(k, tradeoff)
    X = case["X"]

    def body(view):
        lo, hi = shard_rows(X.shape[0], view.world, view.rank)
        eng = PrmfEngine(hi - lo, X.shape[1], k, x_dtype="f32", device=0, nranks=view.world, rank=view.rank, dist_mode=mode)
        eng.set_x(X[lo:hi], norm_x_sq=float(np.sum(X * X)))
        eng.set_library(PathwayLibrary.from_edge_lists(X.shape[1], case["pathways"]))
        eng.set_active(np.arange(k) % len(case["pathways"]))
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        runner = DistributedPrmf(eng, group=view, native=False)
        assert not runner.native and runner.coll is view
        objs, gd = runner.step(U, V, 5, case["gamma"], case["delta"], tradeoff)
        out = (view.gather_rows(torch, U, X.shape[0]), V.cpu().numpy(), objs, gd, eng.sizes.kc)
        eng.close()
        return out

    q.put(LocalRanks(world, timeout_s=300).run(body))


@pytest.mark.parametrize("world,k,mode,tradeoff", [(4, 3, "colshard", None), (4, 16, "replicated", None), (5, 13, "colshard", None),
                                                  (4, 8, "colshard", 0.4)])
def test_host_driven_thread_ranks_on_the_gpu(world, k, mode, tradeoff):
    """The phase calls with the rank group's OWN exchanges in between (prmf_amd.ranks.LocalRankView: device tensors handed
    between the threads, summed in rank order) -- no RCCL, no double: 4 ranks with k = 3 (a rank without a column), 5 ranks
    with 301 samples (61- and 60-row shards), the tradeoff feedback -- against the oracle."""
    import warnings
    from oracle import prmf_oracle as orc
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = _in_worker(_host_driven_worker, (world, k, mode, tradeoff), {}, deadline_s=600)
    U, V, objs, gd, kc = res[0]
    for other in res[1:]:
        assert np.array_equal(other[0], U) and np.array_equal(other[1], V) and other[2] == objs
    assert kc == (k if mode == "replicated" else -(-k // world))
    case = _host_driven_case(k, tradeoff)
    pws = [orc.pathway_from_edges(case["X"].shape[1], e, nodes, w) for (e, w, nodes) in case["pathways"]]
    act = [pws[i % len(pws)] for i in range(k)]
    tr = []
    if tradeoff is None:
        Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], act, n_steps=5, gamma=case["gamma"], delta=case["delta"], trace=tr)
    else:
        Ur, Vr, od, g2, d2 = orc.nmf_manifold_vec_update_tradeoff(case["X"], case["U0"], case["V0"], act, n_steps=5, tradeoff=tradeoff,
                                                                  gamma=case["gamma"], delta=case["delta"], trace=tr)
        assert abs(gd[0] - g2) < 1e-4 * g2
    eu, ev = rel(U, Ur), rel(V, Vr)
    eo = max(relmax([o[key] for o in objs], [t[key] for t in tr]) for key in KEYS)
    report("host_driven_thread_ranks[%d-%d-%s]" % (world, k, mode), rel_U=eu, rel_V=ev, rel_obj=eo)
    assert eu < 1e-4 and ev < 1e-4 and eo < 1e-4


def _medium_pathway_worker(world, mode, q):
    from oracle import prmf_oracle as orc
    from prmf_amd import PathwayLibrary
    from prmf_amd.ranks import LocalRanks
    from prmf_amd.solver import nmf_pathway
    m, n, k, P = 700, 3000, 12, 40
    case = synth.small_case(seed=31, m=m, n=n, k=k, P=P, order=60)
    lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
    np.random.seed(5)                                    # rank 0's draws are the solve's

    def body(view):
        rounds, trace = [], []
        U, V, od = nmf_pathway(case["X"], None, library=lib, gamma=1.0, k_latent=k, max_iter=300, round_log=rounds, trace=trace,
                               dist_mode=mode, group=view)
        l2p = od["latent_to_pathway_data"]
        return U, V, [t["obj"] for t in trace], rounds, [[int(p) for p, _s in l2p[kk]] for kk in sorted(l2p)]

    res = LocalRanks(world).run(body)
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    r_ref, t_ref = [], []
    np.random.seed(5)
    Ur, Vr, odr = orc.nmf_pathway(case["X"], pws, gamma=1.0, k_latent=k, max_iter=300, round_log=r_ref, trace=t_ref)
    mr = odr["latent_to_pathway_data"]
    q.put((res, Ur, Vr, [t["obj"] for t in t_ref], r_ref, [[int(p) for p, _s in mr[kk]] for kk in sorted(mr)]))


@pytest.mark.slow
def test_nmf_pathway_at_medium_size_on_five_ranks_vs_oracle(mock):
    """The sharded outer solver where the goldens do not reach (700 x 3000, k = 12 on FIVE ranks: kc = 3, ranks 4 owns no
    column; 40 candidate pathways, several `restrict` rounds, the matching): every rank returns the same result, and it is
    the oracle's -- the same assignment in every round, the same 200 iterations, the same final map, trace / U / V at 1e-4."""
    res, Ur, Vr, t_ref, r_ref, m_ref = _in_worker(_medium_pathway_worker, (5, "colshard"), {"PRMF_RCCL_LIB": mock}, deadline_s=900)
    U0, V0, tr0, rounds0, map0 = res[0]
    for U, V, tr, rounds, mp_ in res[1:]:
        assert np.array_equal(U0, U) and np.array_equal(V0, V) and tr0 == tr and rounds0 == rounds and map0 == mp_
    assert rounds0 == r_ref and len(tr0) == len(t_ref) and map0 == m_ref
    e_tr, eu, ev = relmax(tr0, t_ref), rel(U0, Ur), rel(V0, Vr)
    report("nmf_pathway_medium_five_ranks", iterations=len(tr0), rounds=len(rounds0), rel_trace=e_tr, rel_U=eu, rel_V=ev)
    assert e_tr < 1e-4 and eu < 1e-4 and ev < 1e-4
