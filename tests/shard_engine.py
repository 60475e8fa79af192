"""CPU stand-in for prmf_amd.solver.PrmfEngine's phase interface, used ONLY by the
gloo tests of prmf_amd.distributed: same buffers and layouts as the C-ABI
(include/prmf_hip.h: q_send[r][j][c], v_send = n*kc floats + 3 float64 extras),
arithmetic from the oracle's formulas in float64/float32 numpy.  It lets the
sharding and exchange logic run with world_size 2 on a machine without a GPU."""
import numpy as np
import torch

from oracle import prmf_oracle as orc


class OracleShardEngine(object):
    def __init__(self, X_local, n, k, pathways, active, norm_x_sq, nranks, rank, dist_mode="replicated"):
        self.X = np.asarray(X_local, dtype=np.float64)
        self.m, self.n, self.k = self.X.shape[0], n, k
        self.nranks, self.rank = nranks, rank
        self.dist_mode = dist_mode if nranks > 1 else "colshard"
        # layout ranks: column blocks of the V update (1 in the replicated mode, include/prmf_hip.h)
        self.lr = 1 if self.dist_mode == "replicated" else nranks
        self.kc = (k + self.lr - 1) // self.lr
        self.col0 = (0 if self.lr == 1 else rank) * self.kc
        self.ncols = max(0, min(self.kc, k - self.col0))
        self.pathways, self.active = pathways, list(active)
        self.norm_x_sq = norm_x_sq
        self.extras_off = (n * self.kc + 1) // 2 * 2
        self._buf = None
        self.records = []

    def comm_buffers(self):
        if self._buf is None:
            nq = self.n * self.kc
            vs = self.extras_off + 8
            b = {"gram": torch.zeros(self.k * self.k, dtype=torch.float64),
                 "q_send": torch.zeros(nq * self.lr, dtype=torch.float32),
                 "v_send": torch.zeros(vs, dtype=torch.float32)}
            if self.lr == 1:
                b["q_recv"], b["v_recv"] = b["q_send"], b["v_send"]
            else:
                b["q_recv"] = torch.zeros(nq, dtype=torch.float32)
                b["v_recv"] = torch.zeros(vs * self.lr, dtype=torch.float32)
            self._buf = b
        return self._buf

    def begin(self, U, V, gamma, delta, tradeoff=None):
        self.U, self.V = U, V            # torch float32 CPU tensors, updated in place
        self.gamma, self.delta, self.tradeoff = gamma, delta, tradeoff
        self.records = []
        self.Gv = V.double().numpy().T @ V.double().numpy()

    def phase_u(self):                   # prmf_runner.py:420-422 on the local rows
        U, V = self.U.double().numpy(), self.V.double().numpy()
        num = self.X @ V
        den = U @ self.Gv + U
        ratio = np.ones_like(num)
        np.divide(num, den, out=ratio, where=den != 0)
        U = U * ratio
        self.U.copy_(torch.from_numpy(U).float())
        Uf = self.U.double().numpy()
        self._buf["gram"].copy_(torch.from_numpy((Uf.T @ Uf).ravel()))

    def phase_q(self):                   # :424, laid out by owner column block
        Q = self.X.T @ self.U.double().numpy()
        qs = self._buf["q_send"].view(self.lr, self.n, self.kc)
        qs.zero_()
        for r in range(self.lr):
            c0 = r * self.kc
            nc = max(0, min(self.kc, self.k - c0))
            if nc:
                qs[r, :, :nc] = torch.from_numpy(Q[:, c0:c0 + nc]).float()

    def phase_v(self):                   # :425-444 for the own columns + objective pieces
        V = self.V.double().numpy()
        Gu = self._buf["gram"].view(self.k, self.k).numpy()
        q = self._buf["q_recv"].view(self.n, self.kc).double().numpy()
        vs = self._buf["v_send"]
        out = vs[:self.n * self.kc].view(self.n, self.kc)
        cross = man = ign = 0.0
        for c in range(self.ncols):
            kk = self.col0 + c
            pw = self.pathways[self.active[kk]]
            v = V[:, kk]
            ig = np.zeros(self.n)
            ig[pw.support] = self.delta * (v[pw.support] + 1.0) ** -2
            num = q[:, c] + (self.gamma * pw.W.dot(v) + ig)
            den = V @ Gu[:, kk] + self.gamma * pw.deg * v
            den[den < orc.EPSILON] = orc.EPSILON
            vn = v * (num / den)
            vn[vn < orc.EPSILON] = orc.EPSILON
            vn = vn.astype(np.float32).astype(np.float64)
            out[:, c] = torch.from_numpy(vn).float()
            cross += float(q[:, c] @ vn)
            vh = vn / np.linalg.norm(vn)
            man += float(pw.L_normal().dot(vh).dot(vh))
            ign += float(np.sum(1.0 / (vh[pw.support] + 1.0)))
        ex = np.array([cross, man, ign, 0.0])
        vs[self.extras_off:self.extras_off + 8] = torch.from_numpy(ex.view(np.float32).copy())

    def phase_post(self):
        vr = self._buf["v_recv"]
        stride = self.extras_off + 8
        cross = man = ign = 0.0
        for r in range(self.lr):
            blk = vr[r * stride:(r + 1) * stride]
            c0 = r * self.kc
            nc = max(0, min(self.kc, self.k - c0))
            if nc:
                self.V[:, c0:c0 + nc] = blk[:self.n * self.kc].view(self.n, self.kc)[:, :nc]
            ex = blk[self.extras_off:self.extras_off + 8].numpy().view(np.float64)
            cross, man, ign = cross + ex[0], man + ex[1], ign + ex[2]
        Vd = self.V.double().numpy()
        self.Gv = Vd.T @ Vd
        Gu = self._buf["gram"].view(self.k, self.k).numpy()
        recon_sq = self.norm_x_sq - 2 * cross + float(np.sum(Gu * self.Gv))
        recon = np.sqrt(max(recon_sq, 0.0))
        fro = float(np.trace(Gu))
        rec = {"recon": recon, "manifold": man, "ignore": ign, "fro": fro, "gamma": self.gamma,
               "delta": self.delta, "obj": recon + self.gamma * man + self.delta * ign + fro}
        self.records.append(rec)
        if self.tradeoff is not None:
            d = self.tradeoff * man
            self.gamma = 1 if d == 0 else (1 - self.tradeoff) * recon / d
            self.delta = self.gamma

    def end(self, n_steps):
        return self.records[-n_steps:], (self.gamma, self.delta)
