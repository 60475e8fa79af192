"""Multi-GPU form of the PRMF iteration: one process per GPU, X row-sharded, the
k latent columns (and their pathway Laplacians) sharded for the V-update, three
collectives per iteration over torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm):

    phase_u   P = X_g V, U_g update, partial U_g^T U_g
        all_reduce(gram)                  k*k float64
    phase_q   Q_g = X_g^T U_g by owner column block
        reduce_scatter(q_send -> q_recv)  n*kc float32 per rank
    phase_v   own-column V update + objective pieces
        all_gather(v_send -> v_recv)      n*kc+8 float32 per rank
    phase_post  replicated V, V^T V, objective record

The reference has no counterpart (single process, script/prmf_runner.py:419-449);
the math of each phase is the same lines split by rows of X / columns of V.

`engine` is anything with the phase interface of prmf_amd.solver.PrmfEngine
(comm_buffers/begin/phase_*/end); tests drive this module over gloo with a
CPU stand-in engine to check the sharding and exchange logic without a GPU.
"""
import torch
import torch.distributed as dist


class Collectives(object):
    """The three exchanges, on whatever backend the process group has.  NCCL/RCCL
    runs them on device buffers directly; gloo (tests) stages through the host and
    builds reduce_scatter from all_reduce (gloo has no reduce_scatter)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.native = dist.get_backend(group) == "nccl"

    def all_reduce_sum(self, t):
        if self.native or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def reduce_scatter_sum(self, out, inp):
        if self.native:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)
            return
        h = inp.cpu() if inp.is_cuda else inp.clone()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        n = out.numel()
        out.copy_(h.view(-1)[self.rank * n:(self.rank + 1) * n].view_as(out))

    def all_gather(self, out, inp):
        if self.native or not inp.is_cuda:
            dist.all_gather_into_tensor(out, inp, group=self.group)
        else:
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, inp.cpu(), group=self.group)
            out.copy_(ho)


class DistributedPrmf(object):
    """Drives one engine per rank through the four phases with the collectives in
    between.  With world size 1 no collective is issued (q_recv / v_recv alias the
    send buffers)."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.coll = Collectives(group) if self.world > 1 else None
        if self.world != engine.nranks:
            raise ValueError("engine was created for %d ranks, process group has %d" % (engine.nranks, self.world))
        self.buf = engine.comm_buffers()

    def step(self, U, V, n_steps, gamma, delta, tradeoff=None):
        e, b, c = self.engine, self.buf, self.coll
        e.begin(U, V, gamma, delta, tradeoff)
        for _ in range(n_steps):
            e.phase_u()
            if c:
                c.all_reduce_sum(b["gram"])
            e.phase_q()
            if c:
                c.reduce_scatter_sum(b["q_recv"], b["q_send"])
            e.phase_v()
            if c:
                c.all_gather(b["v_recv"], b["v_send"])
            e.phase_post()
        return e.end(n_steps)
