"""The sharded iteration behind the drop-in boundary: `nmf_pathway` called by
every rank of a torch.distributed group, and the CLI under the launcher (`--n-gpus`), must reproduce the reference's
own solve -- same per-round pathway assignment, same final map, same trace, U.csv / V.csv / obj.txt -- with X
row-sharded over two ranks.  The ranks share the one GPU of the test box: gloo between the processes, the in-library
step through the RCCL test double (tests/native/mock_rccl.cpp) or the host-driven collectives."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from measure import rel, relmax, report
from test_gpu_distributed import _collect, _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mock(tmp_path):
    mock = str(tmp_path / "libmock_rccl.so")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(ROOT, "tests", "native", "mock_rccl.cpp"), "-o", mock,
                         "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    return mock


def _solver_worker(rank, world, port, casedir, mode, mock, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mock:
        os.environ["PRMF_RCCL_LIB"] = mock
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import networkx as nx
        from prmf_amd import runner
        from prmf_amd.solver import nmf_pathway
        torch.cuda.set_device(0)
        mydir = os.path.join(casedir, "rank%d" % rank)   # every rank writes its own copy of the (identical) input files
        os.makedirs(mydir)
        c = synth.cli_case(mydir, variant=1)
        graphs = [runner.relabel_nodes(nx.read_graphml(fp).to_undirected(), "name") for fp in c["manifolds"]]
        rounds, trace = [], []
        np.random.seed(1 if rank == 0 else 12345)      # only rank 0's stream may matter
        U, V, od = nmf_pathway(c["X"], graphs, nodelist=c["nodelist_ids"], gamma=1.0, k_latent=6, round_log=rounds, trace=trace,
                               dist_mode=mode)
        l2p = od["latent_to_pathway_data"]
        q.put((rank, U, V, [t["obj"] for t in trace], rounds, [int(l2p[k][0][0]) for k in sorted(l2p)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,native", [("colshard", True), ("replicated", True), ("colshard", False)])
def test_sharded_nmf_pathway_vs_reference(golden_dir, tmp_path, mode, native):
    import torch.multiprocessing as mp
    g = np.load(os.path.join(golden_dir, "g6_api.npz"))
    casedir = str(tmp_path / "case")
    os.makedirs(casedir)
    mock = _mock(tmp_path) if native else None
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_solver_worker, args=(r, 2, port, casedir, mode, mock, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q, deadline_s=600)
    (U0, V0, tr0, rounds0, map0), (U1, V1, tr1, rounds1, map1) = res[0], res[1]
    assert np.array_equal(U0, U1) and np.array_equal(V0, V1) and tr0 == tr1 and rounds0 == rounds1 and map0 == map1
    assert U0.shape == g["U"].shape
    e_tr, eu, ev = relmax(tr0, g["trace"][:len(tr0)]), rel(U0, g["U"]), rel(V0, g["V"])
    report("sharded_nmf_pathway[%s-%s]" % (mode, "in-library" if native else "host-driven"), iterations=len(tr0), rel_trace=e_tr,
           rel_U=eu, rel_V=ev)
    assert np.array_equal(np.array(rounds0), g["rounds"])
    assert len(tr0) == len(g["trace"])
    assert map0 == list(g["final_map"])
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
def test_cli_two_ranks_vs_reference(golden_dir, tmp_path, norm):
    """`prmf_runner.py --n-gpus 2`: the launcher starts two ranks (here sharing cuda:0, gloo between them); rank 0 writes
    the reference's files."""
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "g6_%s.npz" % norm))
    c = synth.cli_case(str(tmp_path), variant=1)
    out = str(tmp_path / "out")
    os.makedirs(out)
    env = dict(os.environ, PRMF_DIST_BACKEND="gloo", PRMF_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py"), "--n-gpus", "2", "--data", c["data"],
                        "--manifolds"] + c["manifolds"] + ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out,
                                                           "--delimiter", "\t", "--seed", "1"] + (["--no-normalize"] if norm == "nonorm" else []),
                       cwd=str(tmp_path), capture_output=True, text=True, env=env, timeout=900)
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
    report("cli_two_ranks[%s]" % norm, iterations=len(trace), rel_trace=et, rel_U=eu, rel_V=ev)
    assert len(trace) == len(g["trace"])
    assert mapping == json.loads(str(g["mapping"]))
    assert et < 1e-4 and eu < 1e-4 and ev < 1e-4
    for key in ref:
        assert abs(vals[key] - ref[key]) <= 1e-4 * abs(ref[key]) + 1e-5, key


@pytest.mark.parametrize("norm", ["nonorm", "norm"])
def test_cli_rank_local_npy_input(golden_dir, tmp_path, norm):
    """`--n-gpus 3` with a binary --data: every rank memory-maps the file and converts only its row shard -- and, with the
    quantile normalisation on, only its COLUMN shard of the transform, handed round in sub-blocks (runner.RowSource;
    PRMF_INPUT_BLOCK_BYTES shrinks the sub-blocks so that this 100 x 1000 matrix takes several rounds).  The run must be the
    reference's run of the delimited file (g6 goldens), and no rank may hold more than 1.5 x its share of float64 input."""
    import re
    import pandas as pd
    g = np.load(os.path.join(golden_dir, "g6_%s.npz" % norm))
    c = synth.cli_case(str(tmp_path), variant=1)
    npy = str(tmp_path / "data.npy")
    np.save(npy, c["X"])
    out = str(tmp_path / "out")
    os.makedirs(out)
    env = dict(os.environ, PRMF_DIST_BACKEND="gloo", PRMF_SHARE_GPU="1", PRMF_INPUT_BLOCK_BYTES=str(32 << 10))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "prmf_runner.py"), "--n-gpus", "3", "--data", npy, "--verbose",
                        "--manifolds"] + c["manifolds"] + ["--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out,
                                                           "--seed", "1"] + (["--no-normalize"] if norm == "nonorm" else []),
                       cwd=str(tmp_path), capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    peaks = {int(a): (int(lo), int(hi), float(mb)) for a, lo, hi, mb in
             re.findall(r"rank (\d+): rows \[(\d+), (\d+)\) of 100 x 1000, peak ([0-9.]+) MB", r.stderr)}
    assert sorted(peaks) == [0, 1, 2], r.stderr[-2000:]
    for rank, (lo, hi, mb) in peaks.items():
        share_mb = (hi - lo) * 1000 * 8 / 1e6
        assert mb <= 1.5 * share_mb + 1e-9, (rank, mb, share_mb)
    # "<iteration> <objective>" lines; --verbose also prints "<latent> <pathway>" pairs (two integers)
    lines = [l for l in r.stdout.splitlines() if len(l.split()) == 2 and l.split()[0].isdigit() and "." in l.split()[1]]
    trace = np.array([float(l.split()[1]) for l in lines])
    vals, mapping = _read_obj(os.path.join(out, "obj.txt"))
    U = pd.read_csv(os.path.join(out, "U.csv"))
    V = pd.read_csv(os.path.join(out, "V.csv"), index_col=0)
    eu, ev, et = rel(U.to_numpy(), g["U"]), rel(V.to_numpy(), g["V"]), relmax(trace, g["trace"][:len(trace)])
    report("cli_rank_local_npy[%s]" % norm, iterations=len(trace), rel_trace=et, rel_U=eu, rel_V=ev,
           peak_over_share=max(mb / ((hi - lo) * 8e-3) for lo, hi, mb in peaks.values()))
    assert len(trace) == len(g["trace"]) and mapping == json.loads(str(g["mapping"]))
    assert et < 1e-4 and eu < 1e-4 and ev < 1e-4
