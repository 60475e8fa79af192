#!/usr/bin/env python
"""Random WHOLE solves on 2 .. 8 thread ranks (nmf_pathway(group=view): rank 0's draws broadcast, rows sharded -- sample counts
below the rank count included --, the in-library multi-rank step through the RCCL test double, U gathered) against the same call
on one rank: the same assignment in every round, the same number of iterations, the same final map, every rank the same result,
trace / U / V within 1e-5 of the single-rank solve (the same arithmetic, another summation order).  Three kinds of case are counted
instead of failed, each with its evidence printed: a stopping rule read at its threshold (quotients 1e-7 apart on either side of tol),
an unstable trajectory (the objective jumps by factors between iterations; four records in five still agree), an ill-conditioned
problem (the single-rank solve is itself 10x+ its usual distance from the float64 oracle).  Not part of the test suite; a failure
prints the case and exits non-zero.
    SEED=1 CASES=40 python tools/fuzz_outer_ranks.py"""
import contextlib
import io
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if "PRMF_RCCL_LIB" not in os.environ:      # build the double and re-run with it bound (the library binds its RCCL once per process)
    so = os.path.join(tempfile.mkdtemp(prefix="mock_rccl_"), "libmock_rccl.so")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(ROOT, "tests", "native", "mock_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"],
                        capture_output=True, text=True)
    if cc.returncode != 0:
        sys.exit("mock build failed:\n" + cc.stderr[-2000:])
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, PRMF_RCCL_LIB=so)).returncode)

import numpy as np  # noqa: E402
import synth  # noqa: E402
from prmf_amd import PathwayLibrary  # noqa: E402
from prmf_amd.ranks import LocalRanks  # noqa: E402
from prmf_amd.solver import nmf_pathway  # noqa: E402

rs = np.random.RandomState(int(os.environ.get("SEED", 1)))
ncases = int(os.environ.get("CASES", 40))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def draw():
    kind = rs.randint(4)
    if kind == 0:
        m, n = rs.randint(2, 30), rs.randint(20, 200)
    elif kind == 1:
        m, n = rs.randint(20, 300), rs.randint(500, 4000)
    else:
        m, n = rs.randint(30, 900), rs.randint(60, 2500)
    k = int(rs.choice([1, 2, 3, 4, 6, 6, 8, 12, 16, 24, 40, 70, 96]))
    P = int(rs.choice([1, 2, 3, 6, 10, 20, 40]))
    return dict(m=int(m), n=int(n), k=k, P=P, order=int(min(max(3, n // 3), rs.randint(3, 120))), x_dtype=str(rs.choice(["f32", "bf16"])),
                max_iter=int(rs.choice([10, 30, 60])), modulus=int(rs.choice([10, 5, 3])), gamma=float(rs.choice([1.0, 0.1, 5.0])),
                world=int(rs.randint(2, 9)), mode=str(rs.choice(["colshard", "replicated"])),
                seed=int(rs.randint(1 << 30)), np_seed=int(rs.randint(1 << 30)))


t_start = time.time()
worst = {"trace": 0.0, "U": 0.0, "V": 0.0}
unstable = [0]
at_threshold = [0]
illcond = [0]
for ci in range(ncases):
    c = draw()
    m, n, k = c["m"], c["n"], c["k"]
    case = synth.small_case(seed=c["seed"], m=m, n=n, k=k, P=c["P"], order=c["order"], bf16_x=(c["x_dtype"] == "bf16"))
    X = case["X"]
    lib = PathwayLibrary.from_edge_lists(n, case["pathways"])
    kw = dict(gamma=c["gamma"], k_latent=k, max_iter=c["max_iter"], modulus=c["modulus"], x_dtype=c["x_dtype"])
    tag = " ".join("%s=%s" % kv for kv in c.items())

    def solve(view):
        rounds, trace = [], []
        U, V, od = nmf_pathway(X, None, library=lib, round_log=rounds, trace=trace, group=view,
                               dist_mode=c["mode"] if view is not None else None, **kw)
        l2p = od["latent_to_pathway_data"]
        return U, V, [t["obj"] for t in trace], rounds, [[int(p) for p, _s in l2p[kk]] for kk in sorted(l2p)]

    # (the solver's prints are silenced HERE, around both runs: sys.stdout is process-wide, not per thread)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        np.random.seed(c["np_seed"])
        try:
            one = solve(None)
            err1 = None
        except Exception as e:
            one, err1 = None, e
        np.random.seed(c["np_seed"])      # rank 0's stream is the solve's
        try:
            res = LocalRanks(c["world"], timeout_s=240).run(solve)
            errn = None
        except Exception as e:
            res, errn = None, e
    if err1 is not None or errn is not None:
        same = err1 is not None and errn is not None
        print("%3d %s  one rank raised %r, %d ranks raised %r%s" % (ci, tag, err1, c["world"], errn, "" if same else "   FAIL"), flush=True)
        if not same:
            sys.exit(1)
        continue
    bad = []
    U1, V1, tr1, rounds1, map1 = one
    for r in res[1:]:
        if not (np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2] == res[0][2] and r[3] == res[0][3] and r[4] == res[0][4]):
            bad.append("ranks disagree")
            break
    U, V, tr, rounds, mp = res[0]
    first = next((i for i, (a, b) in enumerate(zip(rounds, rounds1)) if a != b), None)
    if first is not None:
        bad.append("per-round assignment differs in round %d" % first)
    if len(tr) != len(tr1):
        # the stopping rule |d obj| / obj < tol (prmf_runner.py:772-774) read two values a summation order apart: report how close
        # the deciding quotient was to tol in the run that stopped first
        short, long_ = (tr, tr1) if len(tr) < len(tr1) else (tr1, tr)
        q = abs(short[-1] - short[-1 - c["modulus"]]) / short[-1] if len(short) > c["modulus"] else float("nan")
        ql = abs(long_[len(short) - 1] - long_[len(short) - 1 - c["modulus"]]) / long_[len(short) - 1] if len(short) > c["modulus"] else float("nan")
        msg = "iterations %d vs %d (stopping quotient %.6e in the shorter run, %.6e in the other; tol 1e-3)" % (len(tr), len(tr1), q, ql)
        if first is None and abs(q - 1e-3) < 1e-6 and abs(ql - 1e-3) < 1e-6:
            # the rule read AT its threshold: the two runs agree to 1e-7 and land on either side of tol -- as the reference would
            # against any other summation order.  Counted, not failed; the common iterations are compared below.
            at_threshold[0] += 1
            print("%3d %s  stopping rule at its threshold: %s" % (ci, tag, msg), flush=True)
            common = min(len(tr), len(tr1))
            e = float(np.max(np.abs(np.array(tr[:common]) - np.array(tr1[:common])) / np.abs(np.array(tr1[:common]))))
            if not e < 4e-5:
                print("      ... but the common iterations differ by %.1e   FAIL" % e, flush=True)
                sys.exit(1)
            continue
        bad.append(msg)
    if mp != map1:
        bad.append("final map differs")
    errs = {}
    note = ""
    if not bad:
        e_tr = np.abs(np.array(tr) - np.array(tr1)) / np.abs(np.array(tr1)) if tr else np.zeros(1)
        errs = {"trace": float(np.max(e_tr)), "U": rel(U, U1), "V": rel(V, V1)}
        # (k > 64: recon from the Gram identity differs by summation order, 4e-5 on recon: tools/fuzz_ranks.py)
        tol = 4e-5 if k > 64 else 1e-5
        bad += [key for key, v in errs.items() if not v < tol]
        if bad == ["trace"] and errs["trace"] < 1e-3 and float(np.percentile(e_tr, 80)) < tol:
            # an unstable trajectory (tiny problems: the objective jumps by factors of 10 - 40 between iterations, and a record at
            # a spike is steeper in the state than the state's own 1e-7 difference): the final factors agree, the decisions agree,
            # four records in five agree -- counted, not failed
            bad, note = [], "   (unstable trajectory: %d of %d records beyond %.0e, worst %.1e)" % (int(np.sum(e_tr >= tol)), len(e_tr), tol, errs["trace"])
            unstable[0] += 1
            errs["trace"] = float(np.percentile(e_tr, 80))
        if bad and set(bad) <= {"U", "V", "trace"} and max(errs.values()) < 1e-3:
            # Is the PROBLEM ill-conditioned?  (a handful of samples, a nearly empty latent column: one entry of V grows to 1e3 - 1e6
            # times its neighbours and compounds every rounding difference.)  Ask the float64 oracle: when the single-rank solve is
            # itself >= 10x its usual 1e-7 away from it, the two float32 runs differing by the same order says nothing about the
            # rank path -- counted, not failed.  A defect of the rank path shows as a multi-rank error far beyond the single-rank one.
            from oracle import prmf_oracle as orc
            pws = [orc.pathway_from_edges(n, e_, nodes, w) for (e_, w, nodes) in case["pathways"]]
            Xo = X.astype(np.float32).astype(np.float64) if c["x_dtype"] == "f32" else X
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                np.random.seed(c["np_seed"])
                Ur, Vr, _od = orc.nmf_pathway(Xo, pws, gamma=c["gamma"], k_latent=k, max_iter=c["max_iter"], modulus=c["modulus"])
            e1 = max(rel(U1, Ur), rel(V1, Vr))
            en = max(rel(U, Ur), rel(V, Vr))
            if e1 > 1e-6 and en < 30 * e1:
                note = "   (ill-conditioned problem: single rank %.1e from the float64 oracle, %d ranks %.1e)" % (e1, c["world"], en)
                bad = []
                illcond[0] += 1
                errs = {}
        for key, v in errs.items():
            worst[key] = max(worst[key], v)
    print("%3d %s  iterations %d rounds %d  %s%s" % (ci, tag, len(tr), len(rounds), " ".join("%s %.1e" % kv for kv in errs.items()),
                                                     ("   FAIL: " + ", ".join(bad)) if bad else note), flush=True)
    if bad:
        sys.exit(1)
print("all %d solves: every rank the same result, the same decisions as the single-rank solve, trace / U / V within 1e-5 (worst %s; %d unstable trajectories, %d stops at the threshold, %d ill-conditioned problems; %.0f s)"
      % (ncases, " ".join("%s %.1e" % kv for kv in worst.items()), unstable[0], at_threshold[0], illcond[0], time.time() - t_start))
