"""world_size-2 test of the multi-rank iteration (prmf_amd/distributed.py) over
gloo on CPU: row-sharded X; replicated mode (all_reduce of U^T U and of Q) and
column-sharded mode (all_reduce / reduce_scatter / all_gather) between the phases
-- against the single-process oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, tradeoff, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import prmf_oracle as orc
        from prmf_amd.distributed import DistributedPrmf
        from prmf_amd.synthetic import shard_rows
        from shard_engine import OracleShardEngine
        case = synth.small_case(seed=11, m=90, n=300, k=k, P=6, order=30)
        X = case["X"]
        pws = [orc.pathway_from_edges(X.shape[1], e, nodes, w) for (e, w, nodes) in case["pathways"]]
        active = [i % len(pws) for i in range(k)]
        lo, hi = shard_rows(X.shape[0], world, rank)
        eng = OracleShardEngine(X[lo:hi], X.shape[1], k, pws, active, float(np.sum(X * X)), world, rank, dist_mode=mode)
        U = torch.from_numpy(case["U0"][lo:hi]).float().contiguous()
        V = torch.from_numpy(case["V0"]).float().contiguous()
        runner = DistributedPrmf(eng)
        objs, gd = runner.step(U, V, 5, case["gamma"], case["delta"], tradeoff)
        Ug = [torch.zeros(shard_rows(X.shape[0], world, r)[1] - shard_rows(X.shape[0], world, r)[0], k) for r in range(world)]
        dist.all_gather(Ug, U)
        if rank == 0:
            q.put((torch.cat(Ug).numpy(), V.numpy(), objs, gd))
        # every rank must hold the same replicated V and objective
        Vs = [torch.zeros_like(V) for _ in range(world)]
        dist.all_gather(Vs, V)
        assert all(torch.equal(Vs[0], v) for v in Vs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,tradeoff,mode", [(6, None, "replicated"), (5, None, "replicated"), (6, 0.6, "replicated"),
                                             (6, None, "colshard"), (5, None, "colshard"), (6, 0.6, "colshard")])
def test_two_rank_iteration_matches_oracle(k, tradeoff, mode):
    from oracle import prmf_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, tradeoff, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    U, V, objs, gd = q.get()
    case = synth.small_case(seed=11, m=90, n=300, k=k, P=6, order=30)
    pws = [orc.pathway_from_edges(300, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    active = [pws[i % len(pws)] for i in range(k)]
    tr = []
    if tradeoff is None:
        Ur, Vr, od = orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], active, n_steps=5,
                                                 gamma=case["gamma"], delta=case["delta"], trace=tr)
    else:
        Ur, Vr, od, g2, d2 = orc.nmf_manifold_vec_update_tradeoff(case["X"], case["U0"], case["V0"], active, n_steps=5,
                                                                  tradeoff=tradeoff, gamma=case["gamma"],
                                                                  delta=case["delta"], trace=tr)
        assert abs(gd[0] - g2) < 1e-4 * g2
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(U, Ur) < 1e-5 and rel(V, Vr) < 1e-5        # float32 storage of the stand-in
    for a, b in zip(objs, tr):
        for key in ("recon", "manifold", "ignore", "fro", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key
