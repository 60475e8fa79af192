"""Where a column-split epilogue workgroup spends its time: 8 wall-clock stamps per workgroup (100 MHz counter) --
start | G staged | slab sum done | MFMA chain done | barrier passed | update + stores issued | tile exchanged | Gram done."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P = [int(x) for x in os.environ.get("SHAPE", "25000,25000,128,1000").split(",")]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
eng = PrmfEngine(m, n, k, x_dtype="bf16")
eng.set_x(Xd); del Xd
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
U = syn.init_factor_torch(torch, dev, 1, m, k); V = syn.init_factor_torch(torch, dev, 2, n, k)
eng.begin(U, V, 1.0, 1.0)
for i in range(2):
    eng.phase_u(); eng.phase_q(); eng.phase_v(); eng.phase_post()
torch.cuda.synchronize()
st = torch.zeros(8 * 2048, dtype=torch.int64, device=dev)
cs = k > 64
names = (["G staged", "slab sum", "MFMA chain", "barrier", "update+stores", "pack+exchange", "Gram"] if cs else
         ["G staged", "slab + U loads", "MFMA chain", "update+stores", "pack", "Gram round", "-"])
# k <= 64: the V kernel carries stamps only in a variant build (-DPRMF_EPI_V_STAMPS: python -c "from prmf_amd import _build;
# _build.build_variant('vstamps', ['PRMF_EPI_V_STAMPS'])", then PRMF_HIP_LIB=prmf_amd/lib/variants/libprmf_hip_vstamps.so VSTAMPS=1)
vst = os.environ.get("VSTAMPS", "0") == "1"
names_v = names if cs else ["G staged", "slab sum (+ exchange)", "tile load + MFMA chain", "V, positions | pathway terms | update", "pack", "Gram round", "-"]
sides = (("U", None, eng.phase_u), ("V", eng.phase_q, eng.phase_v)) if (cs or vst) else (("U", None, eng.phase_u),)
for side, pre, ph in sides:
    if side == "V":
        names = names_v
    if pre:
        pre()
    torch.cuda.synchronize()
    st.zero_()
    eng.lib.prmf_debug_set_epi_stamps(C.c_void_p(st.data_ptr()))
    ph(); torch.cuda.synchronize()
    eng.lib.prmf_debug_set_epi_stamps(None)
    a = st.cpu().numpy().reshape(-1, 8); a = a[a[:, 7] > 0]
    t0 = a[:, 0].min()
    d = np.diff(a, axis=1) / 100.0
    print("%s epilogue: %d workgroups; start spread p50 %.1f max %.1f us; end of first round p50 %.1f max %.1f us" % (
        side, len(a), np.median(a[:, 0] - t0) / 100, (a[:, 0] - t0).max() / 100, np.median(a[:, 7] - t0) / 100, (a[:, 7] - t0).max() / 100))
    print("   phase medians (us): " + ", ".join("%s %.1f" % (nm, np.median(d[:, j])) for j, nm in enumerate(names)))
    if side == "U":
        eng.phase_q(); eng.phase_v(); eng.phase_post()
try:
    eng.end(3)
except Exception as e:
    print(e)
