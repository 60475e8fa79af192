"""Device time per iteration of ONE rank's share of an N-rank run (no collectives: the exchange buffers are
left as they are, so the numbers are meaningless but the kernel sequence and sizes are the real ones)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m, n, k, P = syn.CONFIGS[os.environ.get("WL", "c3")][:4]        # the global problem the ranks share
dev = "cuda:0"
lo, hi = syn.shard_rows(m, N, 0)
Xd = syn.x_rows_torch(torch, dev, 0, n, k, lo, hi, torch.bfloat16)
MODE = os.environ.get("MODE", "replicated")
eng = PrmfEngine(hi - lo, n, k, x_dtype="bf16", nranks=N, rank=0, dist_mode=MODE)
eng.set_x(Xd, norm_x_sq=1e6); del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k, lo, hi); V = syn.init_factor_torch(torch, dev, 2, n, k)
b = eng.comm_buffers()
if os.environ.get("SWEEP_EVENTS"):
    eng.set_timing(True)
for rep in range(3):
    V.copy_(syn.init_factor_torch(torch, dev, 2, n, k))
    eng.invalidate_factors()
    eng.begin(U, V, 1.0, 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(50):
        eng.phase_u(); eng.phase_q()
        if eng.dist_mode == "replicated":
            b["q_send"].mul_(1.0)                                      # stand-in for the in-place all-reduce of Q
            eng.phase_v()
        else:
            b["q_recv"].copy_(b["q_send"][:b["q_recv"].numel()])      # stand-in for the reduce-scatter
            eng.phase_v()
            b["v_recv"].view(N, -1).copy_(b["v_send"].unsqueeze(0).expand(N, -1))   # stand-in for the all-gather
        eng.phase_post()
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    try:
        eng.end(50)
    except FloatingPointError:
        pass
sw = " sweeps %.1f + %.1f us" % (eng.timing()[0] * 1e3, eng.timing()[1] * 1e3) if os.environ.get("SWEEP_EVENTS") else ""
print(eng.dist_mode, end=" "); print("N=%d rank share %dx%d: %.1f us per iteration (device, incl. two stand-in copies)%s" % (N, hi - lo, n, el / 50 * 1e6, sw))
