"""Can a small kernel on a second stream run NEXT TO the persistent sweep grid (the question behind putting the k x k
all-reduce of U^T U on a side stream beside the X^T U sweep)?  A tiny elementwise kernel is enqueued on a side stream
right after the sweep has been enqueued on the main stream; HIP events give when it started / ended relative to the sweep.
    python tools/side_stream_cosched.py c3 c4share8 c5shard"""
import math
import sys

import torch

sys.path.insert(0, ".")
from prmf_amd import PathwayLibrary, synthetic as syn
from prmf_amd.solver import PrmfEngine

for wl in sys.argv[1:] or ["c3"]:
    m, n, k, P, xd = syn.CONFIGS[wl]
    dev = "cuda:0"
    Xd = syn.x_rows_torch(torch, dev, 0, n, k, 0, m, torch.bfloat16 if xd == "bf16" else torch.float32)
    eng = PrmfEngine(m, n, k, x_dtype=xd, eager_objective=True)
    nsq = eng.set_x(Xd)
    del Xd
    eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, 0)))
    eng.set_active([kk % P for kk in range(k)])
    U = syn.init_factor_torch(torch, dev, 1, m, k)
    V = syn.init_factor_torch(torch, dev, 2, n, k)
    nx = math.sqrt(nsq)
    eng.comm_buffers()
    side = torch.cuda.Stream()
    small = torch.zeros(k * k, dtype=torch.float64, device=dev)          # the size of the Gram message
    res = []
    eng.begin(U, V, nx / k, 10.0 / nx)
    for it in range(8):
        eng.phase_u()
        e0, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s1, s2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main = torch.cuda.current_stream()
        e0.record(main)
        eng.phase_q()                                   # the X^T U sweep (persistent grid, two workgroups per CU)
        e_end.record(main)
        with torch.cuda.stream(side):
            side.wait_event(e0)
            s1.record(side)
            small.add_(1.0)                             # stand-in for the collective's kernel: one small workgroup
            s2.record(side)
        eng.phase_v()
        eng.phase_post()
        torch.cuda.synchronize()
        if it >= 2:
            res.append((e0.elapsed_time(e_end) * 1e3, e0.elapsed_time(s1) * 1e3, e0.elapsed_time(s2) * 1e3))
    eng.end(8)
    a = [sum(x[i] for x in res) / len(res) for i in range(3)]
    print("%-9s sweep (+ dense-Q sum) ends %.0f us after its enqueue point; the side-stream kernel became runnable at %.0f us and "
          "finished at %.0f us -> %s" % (wl, a[0], a[1], a[2], "waited for the sweep" if a[2] > 0.8 * a[0] else "ran beside the sweep"))
    eng.close()
