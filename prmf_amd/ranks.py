"""Who the ranks of a sharded solve are and how they talk on the HOST side.

Two ways to be N ranks on one node, one protocol:

    TorchRanks   one PROCESS per GPU, an initialised torch.distributed group (backend "nccl" = RCCL over xGMI; gloo in the
                 CPU tests) -- what `torch.distributed.run` / `prmf_runner.py --n-gpus N` start.
    LocalRanks   one host THREAD per GPU inside a single process (the single-process multi-device form SURVEY 8e asks to
                 keep): `LocalRanks(N).run(fn)` calls `fn(view)` on N threads, rank r bound to device r % device_count.
                 Python only drives; every library call releases the GIL, the in-library RCCL step runs n_steps
                 iterations per call.  On a box with fewer GPUs than ranks several ranks share a device -- RCCL itself
                 refuses that, the RCCL test double (tests/native/mock_rccl.cpp) does not: the GPU tests run the 8-way
                 partition of BASELINE configs[3-4] on the one test GPU this way (the box allows few processes per card).

The reference has no counterpart (one process, script/prmf_runner.py:556).  The protocol is what nmf_pathway, the
engine's communicator bootstrap and the host-driven collectives need:

    world, rank, device_index
    from_rank0(obj)            rank 0's (small, picklable) host object on every rank
    array_from_rank0(a, shape) rank 0's float64 host array on every rank -- as a tensor, not a pickle
    sum_float(x) / min_int(x) / max_floats(xs)  scalar reductions (norms, collective decisions, timings)
    gather_rows(torch, t, m)   row shards of a device tensor -> the full host matrix on every rank
    barrier()
    all_reduce_sum(t) / reduce_scatter_sum(out, inp) / all_gather(out, inp)   on device tensors (host-driven exchanges)
"""
import copy
import threading
import time

import numpy as np


def shard_of(m, world, rank):
    from .synthetic import shard_rows
    return shard_rows(m, world, rank)


class TorchRanks(object):
    """The ranks of a torch.distributed process group (None: the world; not initialised: one rank)."""

    def __init__(self, group=None):
        import os
        self.group, self.world, self.rank, self.dist = group, 1, 0, None
        self.device_index = None
        try:
            import torch.distributed as dist
        except ImportError:
            return
        # PRMF_FORCE_RANK_PATH=1: an initialised group of ONE rank is driven like any other -- its broadcasts, reductions,
        # the RCCL communicator bootstrap and prmf_step_dist all run (on one rank they are identities).  This is how the code
        # the 8-GPU launch executes runs on a one-GPU box with the real librccl, which refuses two ranks on one device
        # (tests/test_gpu_rank_process.py); nothing else reads the variable.
        forced = bool(os.environ.get("PRMF_FORCE_RANK_PATH"))
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced):
            self.dist, self.world, self.rank = dist, dist.get_world_size(group), dist.get_rank(group)

    @property
    def multi(self):
        return self.dist is not None

    @property
    def on_gpu(self):
        return self.multi and self.dist.get_backend(self.group) == "nccl"

    def _src(self):
        return self.dist.get_global_rank(self.group, 0) if self.group is not None else 0

    def _cdev(self, device):
        return device if self.on_gpu else "cpu"

    def barrier(self):
        if self.multi:
            self.dist.barrier(group=self.group)

    def from_rank0(self, obj):
        if not self.multi:
            return obj
        box = [obj if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=self._src(), group=self.group)
        return box[0]

    def array_from_rank0(self, a, shape, device="cpu"):
        if not self.multi:
            return a
        import torch
        cdev = self._cdev(device)
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(cdev) if self.rank == 0 else \
            torch.empty(tuple(shape), dtype=torch.float64, device=cdev)
        self.dist.broadcast(t, src=self._src(), group=self.group)
        return t.cpu().numpy()

    def array_from_rank(self, src, a, shape, device="cpu"):
        """rank `src`'s float64 host array (shape known to everybody) on every rank, as a tensor broadcast"""
        if not self.multi:
            return a
        import torch
        cdev = self._cdev(device)
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(cdev) if self.rank == src else \
            torch.empty(tuple(shape), dtype=torch.float64, device=cdev)
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
        return t.cpu().numpy()

    def bytes_from_rank0(self, buf, device="cpu"):
        """uint8 numpy array of fixed length, rank 0's content on every rank"""
        if not self.multi:
            return buf
        import torch
        t = torch.from_numpy(buf).to(self._cdev(device))
        self.dist.broadcast(t, src=self._src(), group=self.group)
        return t.cpu().numpy().copy()

    def sum_float(self, value, device="cpu"):
        if not self.multi:
            return float(value)
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=self._cdev(device))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def min_int(self, value, device="cpu"):
        if not self.multi:
            return int(value)
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._cdev(device))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def max_floats(self, values, device="cpu"):
        """element-wise maximum over the ranks of a short list of floats (timings)"""
        if not self.multi:
            return [float(v) for v in values]
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self._cdev(device))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return [float(x) for x in t.tolist()]

    def gather_rows(self, torch, t, m):
        if not self.multi:
            return t.cpu().numpy()
        rows = max(shard_of(m, self.world, r)[1] - shard_of(m, self.world, r)[0] for r in range(self.world))
        pad = torch.zeros((rows, t.shape[1]), dtype=t.dtype, device=t.device if self.on_gpu else "cpu")
        pad[:t.shape[0]] = t if self.on_gpu else t.cpu()
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad, group=self.group)
        parts = []
        for r in range(self.world):
            lo, hi = shard_of(m, self.world, r)
            parts.append(out[r][:hi - lo].cpu().numpy())
        return np.concatenate(parts, axis=0)

    # -- device-tensor exchanges of the host-driven iteration (prmf_amd.distributed) --------------------------------
    # NCCL/RCCL runs them on the device buffers directly; gloo (tests) stages through the host and builds
    # reduce_scatter from all_reduce (gloo has no reduce_scatter)
    def all_reduce_sum(self, t):
        dist = self.dist
        if self.on_gpu or not t.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def reduce_scatter_sum(self, out, inp):
        dist = self.dist
        if self.on_gpu:
            dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)
            return
        h = inp.cpu() if inp.is_cuda else inp.clone()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        n = out.numel()
        out.copy_(h.view(-1)[self.rank * n:(self.rank + 1) * n].view_as(out))

    def all_gather(self, out, inp):
        import torch
        dist = self.dist
        if self.on_gpu or not inp.is_cuda:
            dist.all_gather_into_tensor(out, inp, group=self.group)
        else:
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, inp.cpu(), group=self.group)
            out.copy_(ho)


class LocalRanks(object):
    """N ranks as N threads of this process.  `run(fn, *args)` -> [fn(view_r, *args) for r]; an exception on any rank
    breaks the barrier the others wait at and is re-raised after every thread has come back or ONE overall deadline
    (`timeout_s` for the whole run, not per thread) has passed -- a rank stuck inside a library call cannot be
    interrupted, the library's own bounded waits end it (prmf_set_comm_timeout).  After a failure the object is spent:
    its barrier stays broken and abandoned daemon threads may still hold it -- make a new LocalRanks for the next run."""

    def __init__(self, world, devices=None, timeout_s=900.0):
        if world < 1:
            raise ValueError("world must be >= 1")
        self.world = int(world)
        self.timeout_s = float(timeout_s)
        self._barrier = threading.Barrier(self.world, timeout=self.timeout_s)
        self._slots = [None] * self.world
        self._failed = False
        if devices is None:
            try:
                import torch
                nd = max(1, torch.cuda.device_count())
            except Exception:
                nd = 1
            devices = [r % nd for r in range(self.world)]
        if len(devices) != self.world:
            raise ValueError("one device index per rank")
        self.devices = [int(d) for d in devices]

    def view(self, rank):
        return LocalRankView(self, rank)

    def run(self, fn, *args, **kwargs):
        if self._failed:
            raise RuntimeError("this LocalRanks group has failed before (broken barrier, ranks possibly abandoned): create a new one")
        results = [None] * self.world
        errors = [None] * self.world

        def body(r):
            try:
                view = self.view(r)
                stream = None
                try:
                    import torch
                    if torch.cuda.is_available():
                        torch.cuda.set_device(view.device_index)
                        stream = torch.cuda.Stream(view.device_index)      # a stream of its own per rank (current stream is per thread)
                except ImportError:
                    pass
                if stream is not None:
                    with torch.cuda.stream(stream):
                        results[r] = fn(view, *args, **kwargs)
                        stream.synchronize()
                else:
                    results[r] = fn(view, *args, **kwargs)
            except BaseException as e:      # noqa: B902 -- handed to the caller below
                errors[r] = e
                self._barrier.abort()

        threads = [threading.Thread(target=body, args=(r,), name="prmf-rank-%d" % r, daemon=True) for r in range(self.world)]
        for t in threads:
            t.start()
        deadline = time.monotonic() + self.timeout_s      # one deadline for the whole job, not timeout_s per thread
        for t in threads:
            t.join(max(0.0, deadline - time.monotonic()))
        stuck = [t.name for t in threads if t.is_alive()]
        if stuck or any(e is not None for e in errors):
            self._failed = True
        first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
        if first is None:
            first = next((e for e in errors if e is not None), None)
        if first is not None:
            raise first
        if stuck:
            raise RuntimeError("ranks did not finish within %.0f s: %s" % (self.timeout_s, ", ".join(stuck)))
        return results


class LocalRankView(object):
    """Rank `rank` of a LocalRanks group (handed to the function `LocalRanks.run` calls on that rank's thread)."""

    def __init__(self, owner, rank):
        self.owner, self.rank, self.world = owner, int(rank), owner.world
        self.device_index = owner.devices[rank]
        self.group = self

    @property
    def multi(self):
        return self.world > 1

    on_gpu = True      # the ranks share an address space: device tensors are exchanged as they are

    def _exchange(self, value):
        """every rank's value, in rank order (two barriers: all deposited / all read)"""
        o = self.owner
        o._slots[self.rank] = value
        o._barrier.wait()
        vals = list(o._slots)
        o._barrier.wait()
        return vals

    def barrier(self):
        if self.multi:
            self.owner._barrier.wait()

    # Like the process ranks (pickle / tensor broadcast) every rank gets an object of its OWN: the source rank keeps
    # what it passed in, the others a deep copy taken between the two barriers of the exchange -- rank 0 may overwrite
    # its dict for the next round while a peer still reads (nmf_pathway's k_to_lapl), and a caller's list is not
    # appended to by N threads.
    def _from(self, src, value):
        o = self.owner
        o._slots[self.rank] = value
        o._barrier.wait()
        got = o._slots[src]
        if self.rank != src:
            got = copy.deepcopy(got)
        o._barrier.wait()
        return got

    def from_rank0(self, obj):
        return self._from(0, obj) if self.multi else obj

    def array_from_rank0(self, a, shape, device="cpu"):
        return self._from(0, a) if self.multi else a

    def array_from_rank(self, src, a, shape, device="cpu"):
        return self._from(src, a) if self.multi else a

    def bytes_from_rank0(self, buf, device="cpu"):
        return self._exchange(buf)[0].copy() if self.multi else buf

    def sum_float(self, value, device="cpu"):
        vals = self._exchange(float(value)) if self.multi else [float(value)]
        acc = 0.0
        for v in vals:            # rank order on every rank: the same float on all of them
            acc += v
        return acc

    def min_int(self, value, device="cpu"):
        return min(self._exchange(int(value))) if self.multi else int(value)

    def max_floats(self, values, device="cpu"):
        vals = self._exchange([float(v) for v in values]) if self.multi else [[float(v) for v in values]]
        return [max(col) for col in zip(*vals)]

    def gather_rows(self, torch, t, m):
        if not self.multi:
            return t.cpu().numpy()
        return np.concatenate(self._exchange(t.cpu().numpy()), axis=0)      # .cpu() waits for this rank's stream

    # -- device-tensor exchanges (host-driven iteration).  Every rank first waits for its own stream (its tensor is
    # complete), reads the peers' tensors, and only after everybody has read may anybody overwrite (in-place all-reduce).
    def _sync(self, t):
        if t.is_cuda:
            import torch
            torch.cuda.current_stream(t.device).synchronize()

    def _sum_slices(self, parts, t_like, sl=None):
        acc = None
        for p in parts:
            q = (p if sl is None else p.view(-1)[sl]).to(t_like.device)
            acc = q.clone() if acc is None else acc.add_(q)
        return acc

    def all_reduce_sum(self, t):
        self._sync(t)
        acc = self._sum_slices(self._exchange_keep(t), t)
        self._sync(t)
        self.owner._barrier.wait()
        t.copy_(acc.view_as(t))

    def reduce_scatter_sum(self, out, inp):
        self._sync(inp)
        n = out.numel()
        acc = self._sum_slices(self._exchange_keep(inp), out, slice(self.rank * n, (self.rank + 1) * n))
        self._sync(out)
        self.owner._barrier.wait()
        out.copy_(acc.view_as(out))

    def all_gather(self, out, inp):
        self._sync(inp)
        parts = self._exchange_keep(inp)
        n = inp.numel()
        for r, p in enumerate(parts):
            out.view(-1)[r * n:(r + 1) * n].copy_(p.view(-1).to(out.device))
        self._sync(out)
        self.owner._barrier.wait()

    def _exchange_keep(self, t):
        """like _exchange, but without the closing barrier: the callers above close with one after they have read"""
        o = self.owner
        o._slots[self.rank] = t
        o._barrier.wait()
        return list(o._slots)


def as_ranks(group):
    """None / a torch.distributed ProcessGroup -> TorchRanks; a LocalRankView -> itself."""
    if isinstance(group, (LocalRankView, TorchRanks)):
        return group
    return TorchRanks(group)
