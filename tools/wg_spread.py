"""Finish-time spread of the sweep's persistent workgroups (debug stamps, 100 MHz real-time counter)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P = 50000, 20000, 64, 300
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
eng = PrmfEngine(m, n, k, x_dtype="bf16")
eng.set_x(Xd); del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
eng.comm_buffers()
eng.begin(U, V, 1.0, 1.0)
for i in range(3):
    eng.phase_u(); eng.phase_q(); eng.phase_v(); eng.phase_post()
torch.cuda.synchronize()
st = torch.zeros(2 * 1024, dtype=torch.int64, device=dev)
eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()))
for name, ph in (("X*V", eng.phase_u), ("X^T*U", eng.phase_q)):
    st.zero_(); torch.cuda.synchronize()
    ph(); torch.cuda.synchronize()
    a = st.cpu().numpy().reshape(-1, 2); a = a[a[:, 1] > 0]
    t0 = a[:, 0].min(); start = (a[:, 0] - t0) / 100.0; end = (a[:, 1] - t0) / 100.0   # microseconds
    print("%-6s WGs %d  start spread %.1f us | end: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us | mean dur %.1f" % (
        name, len(a), start.max(), end.min(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), end.max(), (end - start).mean()))
    if name == "X*V":
        eng.phase_v() if False else None
eng.lib.prmf_debug_set_stamps(None)
