"""Multi-GPU form of the PRMF iteration: one process per GPU, X row-sharded,
collectives over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

Column-sharded mode (default; BASELINE.json's partition): the k latent columns and
their pathway Laplacians are sharded for the V-update, three exchanges

    phase_u   P = X_g V, U_g update, partial U_g^T U_g
        all_reduce(gram)                  k*k float64
    phase_q   Q_g = X_g^T U_g by owner column block
        reduce_scatter(q_send -> q_recv)  n*kc float32 per rank
    phase_v   own-column V update, its per-workgroup sums in the block's tail
        all_gather(v_send -> v_recv)      v_send_floats float32 per rank
    phase_post  replicated V, packed V, V^T V partials

Replicated mode (engine created with dist_mode="replicated"): every rank updates all
of V with the fused single-rank kernels, two exchanges

    phase_u   as above
        all_reduce(gram)                  k*k float64
    phase_q   Q_g = X_g^T U_g  (n x k, this rank's rows of X only)
        all_reduce(q_send)                n*k float32, in place
    phase_v   V update (all columns)
    phase_post  (nothing to launch)

When the library issues the collectives itself (native=True: `prmf_step_dist`) the k x k
all-reduce moves behind phase_q, into one RCCL group with the exchange of Q.

The reference has no counterpart (single process, script/prmf_runner.py:419-449);
the math of each phase is the same lines split by rows of X / columns of V.

`engine` is anything with the phase interface of prmf_amd.solver.PrmfEngine
(comm_buffers/begin/phase_*/end); tests drive this module over gloo with a
CPU stand-in engine to check the sharding and exchange logic without a GPU.
Who the ranks are -- processes of a torch.distributed group or threads of one
process -- is prmf_amd.ranks' business.
"""
from .ranks import TorchRanks, as_ranks

# the device-tensor exchanges live with the rank groups (prmf_amd.ranks); the old name stays importable
Collectives = TorchRanks


class DistributedPrmf(object):
    """Drives one engine per rank through the iteration.

    `group`: a torch.distributed process group (None: the world) or a rank view of prmf_amd.ranks.LocalRanks (ranks as
    threads of one process).

    native=True (default when the process group's backend is "nccl", and for thread ranks): the library
    itself issues the RCCL collectives on the stream between its phases
    (`prmf_step_dist`), n_steps iterations per host call.  native=False: the four
    phases are called from here with the group's collectives in between --
    the form the gloo tests exercise.  With world size 1 no collective is issued.

    There is NO silent fallback: if the in-library communicator cannot be created (every rank learns it together,
    `engine.comm_init` decides collectively) the constructor raises on every rank.  A caller that wants the host-driven
    path asks for it with native=False; `PRMF_ALLOW_HOST_COLLECTIVES=1` turns the error into the old warning + fallback."""

    def __init__(self, engine, group=None, native=None):
        import os
        self.engine = engine
        self.ranks = as_ranks(group)
        self.world = self.ranks.world
        if self.world != engine.nranks:
            raise ValueError("engine was created for %d ranks, process group has %d" % (engine.nranks, self.world))
        if native is None:
            native = self.ranks.multi and self.ranks.on_gpu and hasattr(engine, "comm_init")
        self.native = False
        if native:
            # engine.comm_init decides collectively: it either succeeds on every rank or raises on every rank, so no
            # rank ever ends up inside nccl* calls while another one is in the group's own collectives
            try:
                engine.comm_init(group)
                self.native = True
            except Exception as e:
                if not os.environ.get("PRMF_ALLOW_HOST_COLLECTIVES"):
                    raise
                import sys
                sys.stderr.write("prmf_amd: in-library RCCL communicator unavailable (%s); "
                                 "PRMF_ALLOW_HOST_COLLECTIVES is set: using host-driven collectives\n" % e)
        self.coll = self.ranks if (self.ranks.multi and not self.native) else None
        self.buf = None if self.native else engine.comm_buffers()
        if self.coll is not None and hasattr(engine, "set_global_rows"):
            # host-driven: the library cannot all-reduce the sample count itself (prmf_hip.h: prmf_set_global_rows)
            engine.set_global_rows(int(round(self.ranks.sum_float(engine.m, "cuda:%d" % engine.device))))

    def step(self, U, V, n_steps, gamma, delta, tradeoff=None):
        e = self.engine
        if self.native:
            return e.step_dist(U, V, n_steps, gamma, delta, tradeoff)
        b, c = self.buf, self.coll
        if c is not None and tradeoff is not None and not getattr(self, "_warned_tradeoff", False) and hasattr(e, "lib"):
            # the library can only sum the ranks' direct residuals itself (prmf_step_dist); driven from here the feedback
            # gamma = delta = (1 - t) recon / (t manifold) runs on the Gram-identity recon, whose 1e-7 ||X||^2 error the
            # feedback amplifies: say so instead of differing silently from the in-library path
            import warnings
            warnings.warn("prmf_amd: --tradeoff with host-driven collectives uses the Gram-identity reconstruction error in the "
                          "feedback; the in-library RCCL step (native=True) uses the direct residual", RuntimeWarning)
            self._warned_tradeoff = True
        replicated = getattr(e, "dist_mode", "colshard") == "replicated"
        objs_all = []
        gd = (gamma, delta)
        done = 0
        while done < n_steps or (n_steps == 0 and done == 0):
            batch = min(n_steps - done, 128)         # the library keeps <= 256 records between begin and end
            e.begin(U, V, gd[0], gd[1], tradeoff)
            for _ in range(batch):
                e.phase_u()
                if c:
                    c.all_reduce_sum(b["gram"])
                e.phase_q()
                if c and replicated:
                    c.all_reduce_sum(b["q_send"])
                elif c:
                    c.reduce_scatter_sum(b["q_recv"], b["q_send"])
                e.phase_v()
                if c and not replicated:
                    c.all_gather(b["v_recv"], b["v_send"])
                e.phase_post()
            objs, gd = e.end(batch)
            objs_all += objs
            done += batch
            if n_steps == 0:
                break
        return objs_all, gd
