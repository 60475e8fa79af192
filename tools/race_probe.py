import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine
m, n, k, P = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (50000, 20000, 64, 300))]
dev = "cuda:0"
Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16)
eng = PrmfEngine(m, n, k, x_dtype="bf16")
nsq = eng.set_x(Xd)
eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
eng.set_active([kk % P for kk in range(k)])
normx = math.sqrt(nsq)
U0 = syn.init_factor_torch(torch, dev, 1, m, k); V0 = syn.init_factor_torch(torch, dev, 2, n, k)
# reference P = X V in float64 on a row sample via torch
rows = torch.arange(0, m, max(1, m // 4000), device=dev)
Pref = (Xd[rows].double() @ V0.double())
ref_den = (U0[rows].double() @ (V0.double().T @ V0.double())) + U0[rows].double()
Uref = U0[rows].double() * (Pref / ref_den)
outs = []
for rep in range(4):
    U = U0.clone(); V = V0.clone()
    eng.invalidate_factors()
    eng.comm_buffers()
    eng.begin(U, V, normx / k, 10 / normx)
    eng.phase_u()
    torch.cuda.synchronize()
    Uh = U.clone()
    eng.phase_q(); eng.phase_v(); eng.phase_post()
    objs, _ = None, None
    try:
        objs, _ = eng.end(1)
    except FloatingPointError as e:
        print("rep", rep, "FPE", e)
    bad_u = (~torch.isfinite(Uh)).sum().item(); bad_v = (~torch.isfinite(V)).sum().item()
    err_u = ((Uh[rows].double() - Uref).norm() / Uref.norm()).item()
    print("rep", rep, "nonfinite U", bad_u, "V", bad_v, "relerr U(sample)", "%.3e" % err_u, "obj", objs[0]["obj"] if objs else None)
    outs.append((Uh, V.clone()))
for i in range(1, len(outs)):
    du = (outs[i][0] != outs[0][0]).sum().item(); dv = (outs[i][1] != outs[0][1]).sum().item()
    print("rep", i, "differs from rep0 in U:", du, "V:", dv)
    if du:
        idx = (outs[i][0] != outs[0][0]).nonzero()
        print("  first U diffs rows", idx[:5].tolist(), "rows range", idx[:,0].min().item(), idx[:,0].max().item())
