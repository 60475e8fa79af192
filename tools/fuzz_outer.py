#!/usr/bin/env python
"""Random problems through the WHOLE solver a user calls (nmf_pathway, prmf_runner.py:556-792: the random factors, per-round
pathway sampling, candidate scoring, `restrict`, the inner iterations, the final matching, best iterate, stopping rule) against
the oracle's nmf_pathway consuming the same legacy RNG stream: the same assignment in every round, the same number of
iterations, the same final map, trace / U / V at the 1e-4 bar.  Shapes from a handful of samples to a few thousand genes,
k = 1 .. 24, 1 .. 40 candidate pathways (fewer than k included), both storages of X, with and without the tradeoff
feedback, host matrix / device tensor as the input.  Discrete decisions (the percentile cut of `restrict`, the multinomial draws
over candidate scores, the matching) can legitimately flip on a tie of float64 values the float32 path does not resolve: a case
whose first difference is an assignment is printed with both assignments (STRICT=0: counted and carried on; default: exit 1).
With the tradeoff feedback the reference itself usually runs away (gamma 1e39 -> 1e227 within a few iterations, then NaN scores and
a ValueError out of its multinomial draw): there the product must stop with its numeric error.  Not part of the test suite; a
failure prints the case and exits non-zero.
    SEED=1 CASES=40 python tools/fuzz_outer.py"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from oracle import prmf_oracle as orc  # noqa: E402
from prmf_amd import PathwayLibrary  # noqa: E402
from prmf_amd.solver import nmf_pathway  # noqa: E402

rs = np.random.RandomState(int(os.environ.get("SEED", 1)))
ncases = int(os.environ.get("CASES", 40))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def draw():
    kind = rs.randint(4)
    if kind == 0:
        m, n = rs.randint(3, 40), rs.randint(20, 200)
    elif kind == 1:
        m, n = rs.randint(20, 300), rs.randint(500, 5000)
    else:
        m, n = rs.randint(30, 900), rs.randint(60, 2500)
    k = int(rs.choice([1, 2, 3, 4, 6, 6, 8, 12, 16, 24]))
    P = int(rs.choice([1, 2, 3, 6, 10, 20, 40]))
    return dict(m=int(m), n=int(n), k=k, P=P, order=int(min(max(3, n // 3), rs.randint(3, 120))), x_dtype=str(rs.choice(["f32", "bf16"])),
                max_iter=int(rs.choice([20, 50, 120, 300])), modulus=int(rs.choice([10, 10, 5, 3])), gamma=float(rs.choice([1.0, 0.1, 5.0])),
                tradeoff=(float(rs.choice([0.3, 0.5])) if rs.rand() < 0.15 else None), on_gpu=bool(rs.rand() < 0.3),
                seed=int(rs.randint(1 << 30)), np_seed=int(rs.randint(1 << 30)))


t_start = time.time()
worst = {"trace": 0.0, "U": 0.0, "V": 0.0}
ties = 0
for ci in range(ncases):
    c = draw()
    m, n, k = c["m"], c["n"], c["k"]
    case = synth.small_case(seed=c["seed"], m=m, n=n, k=k, P=c["P"], order=c["order"], bf16_x=(c["x_dtype"] == "bf16"))
    X = case["X"]
    if c["x_dtype"] == "f32":
        X = X.astype(np.float32).astype(np.float64)                      # the values the GPU holds
    pws = [orc.pathway_from_edges(n, e, nodes, w) for (e, w, nodes) in case["pathways"]]
    lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
    kw = dict(gamma=c["gamma"], k_latent=k, max_iter=c["max_iter"], modulus=c["modulus"], tradeoff=c["tradeoff"])
    r_ref, t_ref, rounds, trace = [], [], [], []
    tag = " ".join("%s=%s" % kv for kv in c.items())
    np.random.seed(c["np_seed"])
    try:
        Ur, Vr, odr = orc.nmf_pathway(X, pws, round_log=r_ref, trace=t_ref, **kw)
        ref_err = None
    except Exception as e:      # the reference's own failures (e.g. fewer pathways than latent columns) must be the product's too
        ref_err = e
    np.random.seed(c["np_seed"])
    Xin = torch.from_numpy(X).cuda() if c["on_gpu"] else X
    try:
        U, V, od = nmf_pathway(Xin, None, library=lib, x_dtype=c["x_dtype"], round_log=rounds, trace=trace, **kw)
        err = None
    except Exception as e:
        err = e
    if ref_err is None and isinstance(err, FloatingPointError) and max(max(abs(t[key]) for key in ("obj", "gamma", "delta", "recon")) for t in t_ref) > 1e30:
        # the tradeoff feedback runs away in the reference itself (gamma 1e39 -> 1e227 within a few iterations: float64 carries it,
        # float32 cannot): the product must stop with its numeric error, not return numbers
        print("%3d %s  the reference's own iterates leave float32's range (max %.1e): the product raised %r" % (
            ci, tag, max(abs(t["obj"]) for t in t_ref), err), flush=True)
        continue
    if ref_err is not None or err is not None:
        # the same error -- or the reference dying of its own non-finite iterates (`ValueError: Negative dimensions are not allowed`
        # out of the multinomial draw over NaN scores, after a tradeoff run-away) where the product stops with its numeric error
        same = ref_err is not None and err is not None and (type(ref_err) is type(err) or isinstance(err, FloatingPointError))
        print("%3d %s  reference raised %r, product raised %r%s" % (ci, tag, ref_err, err, "" if same else "   FAIL"), flush=True)
        if not same:
            sys.exit(1)
        continue
    first = next((i for i, (a, b) in enumerate(zip(rounds, r_ref)) if a != b), None)
    mp, mr = od["latent_to_pathway_data"], odr["latent_to_pathway_data"]
    final_same = [[int(p) for p, _s in mp[kk]] for kk in sorted(mp)] == [[int(p) for p, _s in mr[kk]] for kk in sorted(mr)]
    tr, tref = np.array([t["obj"] for t in trace]), np.array([t["obj"] for t in t_ref])
    if first is not None or len(tr) != len(tref) or not final_same:
        # up to the first differing round the traces must agree; report how far they ran together
        common = min(len(tr), len(tref), (first if first is not None else 10 ** 9) * c["modulus"])
        e_common = float(np.max(np.abs(tr[:common] - tref[:common]) / np.abs(tref[:common]))) if common else 0.0
        print("%3d %s  DECISION DIFFERS: first differing round %s, iterations %d vs %d, final map same: %s, trace up to there %.1e"
              % (ci, tag, first, len(tr), len(tref), final_same, e_common), flush=True)
        if first is not None:
            print("      product   round %d: %s\n      reference round %d: %s" % (first, rounds[first], first, r_ref[first]), flush=True)
        ties += 1
        if os.environ.get("STRICT", "1") == "1":
            sys.exit(1)
        continue
    errs = {"trace": float(np.max(np.abs(tr - tref) / np.abs(tref))) if len(tr) else 0.0, "U": rel(U, Ur), "V": rel(V, Vr)}
    for key, v in errs.items():
        worst[key] = max(worst[key], v)
    bad = [key for key, v in errs.items() if not v < 1e-4]
    print("%3d %s  iterations %d rounds %d  %s%s" % (ci, tag, len(tr), len(rounds), " ".join("%s %.1e" % kv for kv in errs.items()),
                                                     ("   FAIL: " + ", ".join(bad)) if bad else ""), flush=True)
    if bad:
        sys.exit(1)
print("all %d solves: the same decisions in every round, trace / U / V within 1e-4 (worst %s; %d decision differences; %.0f s)"
      % (ncases, " ".join("%s %.1e" % kv for kv in worst.items()), ties, time.time() - t_start))
