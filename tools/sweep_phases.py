"""Where a sweep wave spends its cycles (needs a library built with -DPRMF_PHASE_PROF=1): per wave, s_memtime sums of
  0 rest of loop (segment prologue/slab stores)   1 issue of the next chunk's global loads   2 fragment reads + MFMAs
  3 barrier after compute                          4 vmcnt waits + LDS writes                 5 barrier after the writes
SHAPE=m,n,k,P XDTYPE=bf16 PRMF_HIP_LIB=... python tools/sweep_phases.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P = [int(x) for x in os.environ.get("SHAPE", "25000,25000,128,1000").split(",")]
XD = os.environ.get("XDTYPE", "bf16")
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if XD == "bf16" else torch.float32)
eng = PrmfEngine(m, n, k, x_dtype=XD, eager_objective=True)
eng.set_x(Xd); del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
eng.comm_buffers()
eng.begin(U, V, 1.0, 1.0)
for i in range(3):
    eng.phase_u(); eng.phase_q(); eng.phase_v(); eng.phase_post()
torch.cuda.synchronize()
_w, _ph, _mx = C.c_int(), C.c_int(), C.c_int()
st = torch.zeros(eng.lib.prmf_debug_stamp_layout(C.byref(_w), C.byref(_ph), C.byref(_mx)), dtype=torch.int64, device=dev)
assert eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()), st.numel()) == 0
names = ["other", "load issue", "reads+MFMA", "barrier 1", "waits+LDS writes", "barrier 2"]
for name, ph in (("X*V", eng.phase_u), ("X^T*U", eng.phase_q)):
    st.zero_(); torch.cuda.synchronize()
    ph(); torch.cuda.synchronize()
    a = st.cpu().numpy()
    se = a[:1024].reshape(-1, 2); dur = (se[:, 1] - se[:, 0]) / 100.0
    p = a[_ph.value:].reshape(_mx.value, 8, 8)[:512, :4, :6].astype(np.float64)      # (4-wave forms; the 8-wave forms fill waves 4..7 too)
    tot = p.sum(axis=2)
    print("%s: kernel WG duration mean %.1f us; wave cycles mean %.0f" % (name, dur[dur > 0].mean(), tot.mean()))
    for i, nm in enumerate(names):
        print("   %-18s mean %8.0f cycles  %5.1f %%   (wave 0: %5.1f %%  wave 3: %5.1f %%)" % (
            nm, p[:, :, i].mean(), 100 * p[:, :, i].sum() / tot.sum(), 100 * p[:, 0, i].sum() / tot[:, 0].sum(), 100 * p[:, 3, i].sum() / tot[:, 3].sum()))
eng.lib.prmf_debug_set_stamps(None, 0)
