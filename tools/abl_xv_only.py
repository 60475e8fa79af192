"""X V sweep timed with only phase_u repeated: the packed factor V never changes, so an ABLATION build whose arithmetic is
wrong still multiplies realistic operands (garbage never reaches the factor the sweep reads).  WORKLOAD=c5shard|c5|..."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prmf_amd import PathwayLibrary, synthetic as syn  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402

wl = os.environ.get("WORKLOAD", "c5shard")
m, n, k, P, xd = syn.CONFIGS[wl]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
eng = PrmfEngine(m, n, k, x_dtype=xd, eager_objective=True)
eng.set_x(Xd)
del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
eng.set_timing(True)
eng.comm_buffers()
U = syn.init_factor_torch(torch, dev, 1, m, k)
V = syn.init_factor_torch(torch, dev, 2, n, k)
try:
    eng.step(U, V, 30, 1.0, 1.0)        # records every timing event once (prmf_end reads the X^T U events too)
except FloatingPointError:
    pass
res = []
for rep in range(3):
    U = syn.init_factor_torch(torch, dev, 1, m, k)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    eng.invalidate_factors()
    eng.begin(U, V, 1.0, 1.0)
    for i in range(20):
        eng.phase_u()
        eng.lib.prmf_debug_bump_cursor(eng.ctx)
    torch.cuda.synchronize()
    try:
        eng.end(20)
    except FloatingPointError:
        pass
    res.append(eng.timing()[0])
print("%s xv-only %s ms" % (wl, " ".join("%.4f" % r for r in res)))
