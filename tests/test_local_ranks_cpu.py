"""Ranks as threads of one process (prmf_amd.ranks.LocalRanks) on CPU: the host-side protocol (rank 0's objects, sums,
row gathers, the three device-tensor exchanges) and the host-driven multi-rank iteration with FOUR and EIGHT ranks
through a stand-in engine -- shard sizes that differ by one, ranks that own no latent column (k = 5 on 4 ranks, k = 6 on
8), a rank count that exceeds the samples -- against the single-process oracle."""
import numpy as np
import pytest
import torch

import synth
from prmf_amd.ranks import LocalRanks, TorchRanks, as_ranks
from prmf_amd.synthetic import shard_rows


def test_protocol_of_thread_ranks():
    def body(view):
        assert as_ranks(view) is view and view.multi and view.world == 5
        obj = view.from_rank0({"seed": 7} if view.rank == 0 else None)
        arr = view.array_from_rank0(np.arange(6.0).reshape(2, 3) if view.rank == 0 else None, (2, 3))
        total = view.sum_float(0.1 * (view.rank + 1))
        low = view.min_int(10 - view.rank)
        lo, hi = shard_rows(13, view.world, view.rank)
        rows = view.gather_rows(torch, torch.arange(13.0).view(13, 1)[lo:hi], 13)
        t = torch.full((4,), float(view.rank + 1))
        view.all_reduce_sum(t)
        out = torch.zeros(2)
        view.reduce_scatter_sum(out, torch.arange(10.0) * (view.rank + 1))
        gathered = torch.zeros(10)
        view.all_gather(gathered, torch.tensor([view.rank, -view.rank], dtype=torch.float32))
        return obj, arr, total, low, rows, t, out, gathered

    res = LocalRanks(5).run(body)
    ref_total = 0.0
    for r in range(5):
        ref_total += 0.1 * (r + 1)
    for r, (obj, arr, total, low, rows, t, out, gathered) in enumerate(res):
        assert obj == {"seed": 7} and np.array_equal(arr, np.arange(6.0).reshape(2, 3))
        assert total == ref_total and low == 6                      # the same float on every rank (rank-order sum)
        assert np.array_equal(rows[:, 0], np.arange(13.0))
        assert torch.equal(t, torch.full((4,), 15.0))
        assert torch.equal(out, torch.arange(10.0)[2 * r:2 * r + 2] * 15.0)
        assert torch.equal(gathered, torch.tensor([0, 0, 1, -1, 2, -2, 3, -3, 4, -4], dtype=torch.float32))


def test_failure_on_one_rank_reaches_the_caller():
    def body(view):
        if view.rank == 2:
            raise ValueError("rank 2 failed")
        view.barrier()                                              # the others wait here: the barrier is broken for them
        return view.rank

    with pytest.raises(ValueError, match="rank 2 failed"):
        LocalRanks(4, timeout_s=30).run(body)


def test_failed_group_is_spent_and_joins_share_one_deadline():
    import time
    g = LocalRanks(3, timeout_s=30)
    with pytest.raises(ValueError):
        g.run(lambda view: (_ for _ in ()).throw(ValueError("boom")) if view.rank == 1 else view.barrier())
    with pytest.raises(RuntimeError, match="failed before"):
        g.run(lambda view: view.rank)
    # ONE deadline for the whole job: three ranks that never come back are reported after ~timeout_s, not 3 x timeout_s
    stuck = LocalRanks(3, timeout_s=0.4)
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="did not finish"):
        stuck.run(lambda view: time.sleep(3.0))
    assert time.monotonic() - t0 < 1.2


def test_rank0_objects_arrive_as_per_rank_copies():
    """what rank 0 hands out is a COPY on the other ranks (as between processes): rank 0 may go on writing to its own"""
    def body(view):
        d = view.from_rank0({"a": [1, 2]} if view.rank == 0 else None)
        a = view.array_from_rank0(np.zeros(3) if view.rank == 0 else None, (3,))
        view.barrier()
        if view.rank == 0:
            d["a"].append(3)
            d["b"] = 1
            a[:] = 7.0
        view.barrier()
        return d, a

    res = LocalRanks(4).run(body)
    assert res[0][0] == {"a": [1, 2, 3], "b": 1} and np.all(res[0][1] == 7.0)
    for d, a in res[1:]:
        assert d == {"a": [1, 2]} and np.all(a == 0.0)
    assert len({id(d) for d, _a in res}) == 4 and len({id(a) for _d, a in res}) == 4


def test_single_rank_group_is_inert():
    r = TorchRanks(None)
    assert not r.multi and r.from_rank0(3) == 3 and r.sum_float(1.5) == 1.5 and r.min_int(4) == 4
    assert LocalRanks(1).run(lambda v: (v.multi, v.from_rank0("x"), v.sum_float(2.0))) == [(False, "x", 2.0)]


@pytest.mark.parametrize("world,m,k,tradeoff,mode", [(4, 90, 5, None, "colshard"), (4, 90, 6, None, "replicated"), (8, 100, 6, None, "colshard"),
                                                    (8, 100, 6, 0.6, "colshard"), (8, 5, 6, None, "colshard"), (8, 100, 6, None, "replicated")])
def test_many_rank_iteration_matches_oracle(world, m, k, tradeoff, mode):
    from oracle import prmf_oracle as orc
    from prmf_amd.distributed import DistributedPrmf
    from shard_engine import OracleShardEngine
    case = synth.small_case(seed=11, m=m, n=300, k=k, P=6, order=30)
    X = case["X"]
    pws = [orc.pathway_from_edges(X.shape[1], e, nodes, w) for (e, w, nodes) in case["pathways"]]
    active = [i % len(pws) for i in range(k)]

    def body(view):
        lo, hi = shard_rows(X.shape[0], view.world, view.rank)
        eng = OracleShardEngine(X[lo:hi], X.shape[1], k, pws, active, float(np.sum(X * X)), view.world, view.rank, dist_mode=mode)
        U = torch.from_numpy(case["U0"][lo:hi]).float().contiguous()
        V = torch.from_numpy(case["V0"]).float().contiguous()
        objs, gd = DistributedPrmf(eng, group=view, native=False).step(U, V, 5, case["gamma"], case["delta"], tradeoff)
        return view.gather_rows(torch, U, X.shape[0]), V.numpy(), objs, gd, eng.ncols

    res = LocalRanks(world).run(body)
    U, V, objs, gd, _ = res[0]
    for other in res[1:]:
        assert np.array_equal(other[1], V) and other[2] == objs       # replicated state agrees bitwise
    if mode == "colshard":
        assert sum(r[4] for r in res) == k and (k >= world or min(r[4] for r in res) == 0)
    tr = []
    act = [pws[i] for i in active]
    if tradeoff is None:
        Ur, Vr, od = orc.nmf_manifold_vec_update(X, case["U0"], case["V0"], act, n_steps=5, gamma=case["gamma"], delta=case["delta"], trace=tr)
    else:
        Ur, Vr, od, g2, d2 = orc.nmf_manifold_vec_update_tradeoff(X, case["U0"], case["V0"], act, n_steps=5, tradeoff=tradeoff,
                                                                  gamma=case["gamma"], delta=case["delta"], trace=tr)
        assert abs(gd[0] - g2) < 1e-4 * g2
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(U, Ur) < 1e-5 and rel(V, Vr) < 1e-5
    for a, b in zip(objs, tr):
        for key in ("recon", "manifold", "ignore", "fro", "obj"):
            assert abs(a[key] - b[key]) <= 1e-4 * max(abs(b[key]), 1e-12), key
