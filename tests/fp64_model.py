"""Float64 restatement of the PRMF iteration with torch on the GPU -- TEST INFRASTRUCTURE for the sizes the NumPy
oracle cannot finish in seconds (50000 x 20000 and up).  Same formulas as oracle/prmf_oracle.py:update_step and
nmf_manifold_vec_obj (script/prmf_runner.py:420-444, :336-372, :56-65), float64 matmuls in row
chunks, the k per-column pathway graphs flattened into one index list so that the sparse terms of all columns are
two index_add_ calls.  Not the HIP path, and never imported by the product.

test_fp64_model_equals_oracle (tests/test_gpu_fullsize.py) pins it to the oracle at a size both can run.
"""
import math

import numpy as np

EPS = float(np.finfo(np.float32).eps)      # prmf_runner.py:23


class Fp64Model(object):
    def __init__(self, torch, Xd, pathways, active, gamma, delta, chunk=4096):
        """Xd: the stored X (torch, bf16 or f32, on the GPU); pathways: [(edges, weights, nodes)]; active[kk] = pathway
        of latent column kk."""
        self.torch, self.X, self.chunk = torch, Xd, chunk
        self.m, self.n = Xd.shape
        self.k = k = len(active)
        self.gamma, self.delta = float(gamma), float(delta)
        dev = Xd.device
        out_i, in_i, ws, sup_i, self_i, self_w = [], [], [], [], [], []
        for kk, p in enumerate(active):
            edges, w, nodes = pathways[p]
            edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
            w = np.asarray(w, dtype=np.float64)
            u, v = edges[:, 0], edges[:, 1]
            off = u != v
            # W as networkx-1.11 adjacency_matrix builds it (:679): symmetric, a self-loop counted once
            out_i.append(np.concatenate([u, v[off]]) * k + kk)
            in_i.append(np.concatenate([v, u[off]]) * k + kk)
            ws.append(np.concatenate([w, w[off]]))
            sup_i.append(np.asarray(nodes, dtype=np.int64) * k + kk)
            self_i.append(u[~off] * k + kk)
            self_w.append(w[~off])
        t = lambda a, dt: torch.as_tensor(np.concatenate(a), dtype=dt, device=dev)
        self.out_i, self.in_i = t(out_i, torch.long), t(in_i, torch.long)
        self.w = t(ws, torch.float64)
        self.sup_i = t(sup_i, torch.long)
        n = self.n
        deg = torch.zeros(n * k, dtype=torch.float64, device=dev)
        deg.index_add_(0, self.out_i, self.w)                       # column sums of W (:680-682)
        wself = torch.zeros(n * k, dtype=torch.float64, device=dev)
        if sum(len(a) for a in self_i):
            wself.index_add_(0, t(self_i, torch.long), t(self_w, torch.float64))
        self.deg = deg.view(n, k)
        ldiag = deg - wself                                         # diagonal of L = D - W
        mask = torch.zeros(n * k, dtype=torch.bool, device=dev)
        mask[self.sup_i] = True
        self.mask = mask.view(n, k)
        dis = torch.zeros(n * k, dtype=torch.float64, device=dev)   # normalize_laplacian (:56-65): L_ii^-1/2 on the support
        sel = mask & (ldiag != 0)
        dis[sel] = ldiag[sel] ** -0.5
        self.dis = dis.view(n, k)

    def w_times(self, Y):
        """[W_k y_k]_k for an n x k matrix of columns."""
        out = self.torch.zeros(self.n * self.k, dtype=self.torch.float64, device=Y.device)
        out.index_add_(0, self.out_i, self.w * Y.reshape(-1)[self.in_i])
        return out.view(self.n, self.k)

    def xv(self, V):
        torch = self.torch
        P = torch.empty((self.m, self.k), dtype=torch.float64, device=V.device)
        for r0 in range(0, self.m, self.chunk):
            P[r0:r0 + self.chunk] = self.X[r0:r0 + self.chunk].double() @ V
        return P

    def xtu(self, U):
        torch = self.torch
        Q = torch.zeros((self.n, self.k), dtype=torch.float64, device=U.device)
        for r0 in range(0, self.m, self.chunk):
            Q += self.X[r0:r0 + self.chunk].double().T @ U[r0:r0 + self.chunk]
        return Q

    def step(self, U, V):
        """One iteration (:420-444) on float64 U, V.  Returns P = X V, U_new, Q = X^T U_new, V_new."""
        torch = self.torch
        P = self.xv(V)                                                             # :420
        den = U @ (V.T @ V) + U                                                    # :421
        U_new = U * torch.where(den != 0, P / den, torch.ones_like(den))           # :422
        Q = self.xtu(U_new)                                                        # :424
        den_v = V @ (U_new.T @ U_new)                                              # :425
        ign = torch.where(self.mask, self.delta * (V + 1.0) ** -2, torch.zeros_like(V))     # :436-438
        num = Q + (self.gamma * self.w_times(V) + ign)                             # association of :440
        den_v = den_v + self.gamma * (self.deg * V)                                # :435, :441
        den_v = torch.clamp(den_v, min=EPS)                                        # :442
        V_new = torch.clamp(V * (num / den_v), min=EPS)                            # :443-444
        return P, U_new, Q, V_new

    def objective(self, U, V):
        """The record of :336-372 with the materialised residual of :337 (row chunks)."""
        torch = self.torch
        acc = 0.0
        for r0 in range(0, self.m, self.chunk):
            R = self.X[r0:r0 + self.chunk].double() - U[r0:r0 + self.chunk] @ V.T
            acc += float((R * R).sum())
        recon = math.sqrt(acc)
        vhat = V / torch.linalg.norm(V, dim=0, keepdim=True)                       # :345-346
        y = vhat * self.dis
        manifold = float((y * (self.deg * y - self.w_times(y))).sum())             # vhat^T Lhat vhat (:350)
        ignore = float(torch.where(self.mask, 1.0 / (vhat + 1.0), torch.zeros_like(V)).sum())   # :352
        fro = float((U * U).sum())
        return {"recon": recon, "manifold": manifold, "ignore": ignore, "fro": fro, "gamma": self.gamma, "delta": self.delta,
                "obj": recon + self.gamma * manifold + self.delta * ignore + fro}
