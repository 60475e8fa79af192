#!/usr/bin/env python
"""One case of tools/fuzz_ranks.py by its parameters, with the details the fuzzer does not print: the records of the
single-rank run and of every rank, the flags, where a non-finite value first appears.
    CASE="m=3 n=60 k=96 P=1 order=30 x_dtype=f32 steps=4 seed=257628924 world=5 mode=colshard tradeoff=None" python tools/fuzz_ranks_case.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if "PRMF_RCCL_LIB" not in os.environ:
    so = "/tmp/libmock_rccl_case.so"
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(ROOT, "tests", "native", "mock_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, PRMF_RCCL_LIB=so)).returncode)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from oracle import prmf_oracle as orc  # noqa: E402
from prmf_amd import PathwayLibrary  # noqa: E402
from prmf_amd.distributed import DistributedPrmf  # noqa: E402
from prmf_amd.ranks import LocalRanks  # noqa: E402
from prmf_amd.solver import PrmfEngine  # noqa: E402
from prmf_amd.synthetic import shard_rows  # noqa: E402

c = {}
for kv in os.environ["CASE"].split():
    a, b = kv.split("=")
    c[a] = b if a in ("x_dtype", "mode") else (None if b == "None" else (float(b) if a == "tradeoff" else int(b)))
case = synth.small_case(seed=c["seed"], m=c["m"], n=c["n"], k=c["k"], P=c["P"], order=c["order"], zero_genes=min(3, c["n"] // 4),
                        isolated=min(2, max(0, c["order"] - 2)), bf16_x=(c["x_dtype"] == "bf16"))
m, n, k = c["m"], c["n"], c["k"]
lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
active = list(np.arange(k) % len(case["pathways"]))
pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
tr = []
try:
    with np.errstate(all="ignore"):
        orc.nmf_manifold_vec_update(case["X"], case["U0"], case["V0"], [pws[i % len(pws)] for i in range(k)], n_steps=c["steps"],
                                    gamma=case["gamma"], delta=case["delta"], trace=tr)
    print("oracle (float64):", [("%.6g" % t["obj"]) for t in tr], "recon", [("%.6g" % t["recon"]) for t in tr])
except Exception as e:
    print("oracle raised", type(e).__name__, e)


def solve(view):
    world, rank = (view.world, view.rank) if view is not None else (1, 0)
    lo, hi = shard_rows(m, world, rank)
    eng = PrmfEngine(hi - lo, n, k, x_dtype=c["x_dtype"], device=0, nranks=world, rank=rank, dist_mode=c["mode"] if view is not None else None)
    try:
        Xd = eng.device_tensor(case["X"][lo:hi], eng.torch_x_dtype)
        local = eng.local_norm_sq(Xd)
        nsq = view.sum_float(local) if view is not None else local
        eng.set_x(Xd, norm_x_sq=nsq)
        eng.set_library(lib)
        eng.set_active(active)
        stepper = DistributedPrmf(eng, group=view, native=True).step if view is not None else eng.step
        out = []
        U = eng.device_tensor(case["U0"][lo:hi], torch.float32)
        V = eng.device_tensor(case["V0"], torch.float32)
        for s in range(c["steps"]):
            try:
                objs, gd = stepper(U, V, 1, case["gamma"], case["delta"], c["tradeoff"])
                out.append((objs[0]["obj"], objs[0]["recon"], objs[0]["manifold"], objs[0]["fro"], int(eng.last_flags[0]), float(U.abs().max()) if hi > lo else 0.0,
                            float(V.max()), bool(torch.isfinite(U).all()), bool(torch.isfinite(V).all()), int(eng.sweep_info().reserved)))
            except FloatingPointError as e:
                out.append(("numeric", str(e), float(U.abs().max()) if hi > lo else 0.0, float(V.max()), bool(torch.isfinite(U).all()), bool(torch.isfinite(V).all())))
                break
        return out
    finally:
        eng.close()


for label, env in (("default", None), ("PRMF_SWEEP_NO_EXACT=1", "1")):
    if env:
        os.environ["PRMF_SWEEP_NO_EXACT"] = env
    else:
        os.environ.pop("PRMF_SWEEP_NO_EXACT", None)
    print("== %s" % label)
    one = solve(None)
    for s, r in enumerate(one):
        print("  one rank, step %d: %s" % (s, r))
    res = LocalRanks(c["world"], timeout_s=240).run(solve)
    for rk, rr in enumerate(res):
        print("  rank %d of %d: %s" % (rk, c["world"], rr[-1] if rr else None))
