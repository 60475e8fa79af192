import ctypes as C, math, sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, synth
import networkx as nx
from prmf_amd import PathwayLibrary, runner
from prmf_amd.solver import PrmfEngine
c = synth.cli_case("/tmp/case1", variant=1) if os.makedirs("/tmp/case1", exist_ok=True) is None else None
graphs = [runner.relabel_nodes(nx.read_graphml(fp).to_undirected(), "name") for fp in c["manifolds"]]
X = c["X"]; m, n = X.shape; k = 6
lib = PathwayLibrary.from_graphs(graphs, c["nodelist_ids"])
for xd in ("f32",):
    eng = PrmfEngine(m, n, k, x_dtype=xd)
    nsq = eng.set_x(X); eng.set_library(lib); eng.set_active(np.arange(k) % lib.P)
    np.random.seed(1)
    U = eng.device_tensor(3 * (1 - np.random.rand(m, k)), torch.float32); V = eng.device_tensor(3 * (1 - np.random.rand(n, k)), torch.float32)
    nx_ = math.sqrt(nsq)
    for rnd in range(4):
        steps = 10
        objs, _ = eng.step(U, V, steps, nx_ / k, 10 / nx_)
        cross = C.c_double(); gu, gv = np.zeros((k, k)), np.zeros((k, k))
        assert eng.lib.prmf_debug_recon_terms(eng.ctx, steps - 1, C.byref(cross), C.c_void_p(gu.ctypes.data), C.c_void_p(gv.ctypes.data)) == 0
        Xd = torch.as_tensor(X, dtype=torch.float32, device="cuda").double(); U64, V64 = U.double(), V.double()
        acc = float(((Xd - U64 @ V64.T) ** 2).sum()); cross64 = float(((Xd.T @ U64) * V64).sum())
        gu64, gv64 = (U64.T @ U64).cpu().numpy(), (V64.T @ V64).cpu().numpy()
        gg, gg64 = float((gu * gv).sum()), float((gu64 * gv64).sum())
        ident = nsq - 2 * cross.value + gg
        print("round %d recon/||X|| %.4f record recon err %.2e | cross err %.2e | gg err %.2e | Gu max %.2e Gv max %.2e | nsq err %.2e" % (
            rnd, math.sqrt(acc) / nx_, (objs[-1]["recon"] - math.sqrt(acc)) / math.sqrt(acc), 2 * (cross.value - cross64) / nsq, (gg - gg64) / nsq,
            np.abs(gu - gu64).max() / np.abs(gu64).max(), np.abs(gv - gv64).max() / np.abs(gv64).max(), (nsq - float((Xd ** 2).sum())) / nsq))
    eng.close()
