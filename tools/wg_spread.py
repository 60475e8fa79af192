"""Finish-time spread of the sweep's persistent workgroups (debug stamps, 100 MHz real-time counter)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P = [int(x) for x in os.environ.get("SHAPE", "50000,20000,64,300").split(",")]
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
st = torch.zeros(eng.lib.prmf_debug_stamp_layout(C.byref(_w), C.byref(_ph), C.byref(_mx)), dtype=torch.int64, device=dev)   # [0, 2 W): start/end stamps; [_w, _w + W): HW ids
eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()), st.numel())
for name, ph in (("X*V", eng.phase_u), ("X^T*U", eng.phase_q)):
    st.zero_(); torch.cuda.synchronize()
    ph(); torch.cuda.synchronize()
    a = st.cpu().numpy()[:2048].reshape(-1, 2); a = a[a[:, 1] > 0]
    t0 = a[:, 0].min(); start = (a[:, 0] - t0) / 100.0; end = (a[:, 1] - t0) / 100.0   # microseconds
    print("%-6s WGs %d  start: p50 %.1f p90 %.1f max %.1f us | end: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us | mean dur %.1f" % (
        name, len(a), np.percentile(start, 50), np.percentile(start, 90), start.max(), end.min(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), end.max(), (end - start).mean()))
    if name == "X*V":
        eng.phase_v() if False else None
eng.lib.prmf_debug_set_stamps(None, 0)

if os.environ.get('QUICK'):
    sys.exit(0)
# --- are the slow workgroups the same ones every launch? (systematic => a weighted static split would fix it)
import numpy as np
durs = []
eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()), st.numel())
for rep in range(4):
    st.zero_(); torch.cuda.synchronize()
    eng.phase_u(); torch.cuda.synchronize()
    a = st.cpu().numpy()[:2048].reshape(-1, 2)[:512]
    durs.append((a[:, 1] - a[:, 0]) / 100.0)
    eng.phase_q(); eng.phase_v(); eng.phase_post(); torch.cuda.synchronize()
eng.lib.prmf_debug_set_stamps(None, 0)
durs = np.array(durs)
if os.environ.get("SAVE"):
    np.save(os.environ["SAVE"] + "_xv.npy", durs)
print("X*V duration per WG: mean %.1f std %.1f us" % (durs.mean(), durs.std()))
print("corr(launch0, launch1) = %.3f, corr(launch0, launch3) = %.3f" % (np.corrcoef(durs[0], durs[1])[0, 1], np.corrcoef(durs[0], durs[3])[0, 1]))
m = durs.mean(axis=0)
print("per-XCD mean duration (block id %% 8):", np.round([m[x::8].mean() for x in range(8)], 1))
print("by position in XCD (block id // 8) deciles:", np.round([m[np.arange(512) // 8 // 8 == d].mean() for d in range(8)], 1))

eng.lib.prmf_debug_set_stamps(C.c_void_p(st.data_ptr()), st.numel())
durs = []
for rep in range(3):
    eng.phase_u(); torch.cuda.synchronize()
    st.zero_(); torch.cuda.synchronize()
    eng.phase_q(); torch.cuda.synchronize()
    a = st.cpu().numpy()[:2048].reshape(-1, 2)[:512]
    durs.append((a[:, 1] - a[:, 0]) / 100.0)
    eng.phase_v(); eng.phase_post(); torch.cuda.synchronize()
eng.lib.prmf_debug_set_stamps(None, 0)
durs = np.array(durs); m = durs.mean(axis=0)
if os.environ.get("SAVE"):
    np.save(os.environ["SAVE"] + "_xtu.npy", durs)
print("X^T*U duration per WG: mean %.1f std %.1f us; corr %.3f" % (durs.mean(), durs.std(), np.corrcoef(durs[0], durs[2])[0, 1]))
print("per-XCD mean (block id %% 8):", np.round([m[x::8].mean() for x in range(8)], 1))
print("per-XCD, first/second WG per CU:", [(round(m[x::8][:32].mean(), 1), round(m[x::8][32:].mean(), 1)) for x in range(8)])
