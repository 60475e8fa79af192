#!/usr/bin/env python
"""Benchmark of the PRMF inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5]

One "step" = one iteration of the hot path (U-update + V-update + the objective
record, script/prmf_runner.py:420-447) on synthetic inputs of the configuration
BASELINE.json's target is quoted on (c3: X 50000 x 20000 bf16, k = 64, 300 tree
pathways).  X, U, V are resident in HBM before the timed
region.  With N > 1 the SAME global problem is row-sharded over the ranks (strong scaling); per iteration the
k x k Gram U^T U is all-reduced together with the reduce-scatter of the n x k product X^T U by owner column block (one RCCL
group), the V update is sharded by latent column and V is all-gathered (--dist-mode colshard, the default everywhere), or
X^T U is all-reduced and every rank updates all of V (--dist-mode replicated); RCCL issued inside the library.  Rank 0
prints one JSON line.

How N ranks come to be: launched by `torch.distributed.run` (WORLD_SIZE in the environment: one PROCESS per GPU -- what
the driver does), or, started plainly with --gpus N, as N host THREADS of this process, rank r on device r mod
device_count (prmf_amd.ranks.LocalRanks).  On a one-GPU box the thread form with the RCCL test double
(PRMF_RCCL_LIB=libmock_rccl.so) is the rehearsal of the N = 8 run: everything but the wire.
"""
import argparse
import contextlib
import io
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c3f32", "c5", "c5shard", "c5shardf32"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the measurements next to the headline: pure-stream calibration, update-only rate, outer loop")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--cpu-baseline-size", default="auto", choices=["auto", "full", "sample"],
                    help="CPU baseline at the full matrix, on the bounded row sample (extrapolated), or full when the host memory allows (auto)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed --steps block (median reported)")
    ap.add_argument("--eager-objective", action="store_true",
                    help="A/B switch: evaluate the objective every iteration (7 launches per iteration, the round-1 sequence) "
                         "instead of the default 4-launch iteration with the records evaluated per batch")
    ap.add_argument("--dist-mode", default="colshard", choices=["colshard", "replicated"],
                    help="N > 1: how the V update is shared.  colshard (default; BASELINE.json's wording): the k latent "
                         "columns and their pathway Laplacians are sharded -- reduce-scatter of X^T U by owner column "
                         "block, own-column V update, all-gather of V.  replicated: all-reduce of X^T U, every rank "
                         "updates all of V.")
    return ap.parse_args()


def host_info():
    """CPU model, thread count, BLAS vendor/threads and numpy/scipy versions of the box the CPU baseline ran on."""
    import platform
    import scipy
    info = {"os_cpu_count": os.cpu_count(), "numpy": np.__version__, "scipy": scipy.__version__,
            "python": platform.python_version()}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        from threadpoolctl import threadpool_info
        info["blas"] = [{k: p.get(k) for k in ("user_api", "internal_api", "version", "num_threads", "threading_layer")}
                        for p in threadpool_info()]
    except Exception:
        info["blas"] = None
    info["mem_available_gb"] = mem_available_gb()
    return info


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return None


def cpu_baseline(torch, Xd, U0, V0, pathways, active, gamma, delta, m_full, budget_s, size="auto"):
    """The oracle (float64 NumPy/SciPy restatement of prmf_runner.py:419-447) timed on this box's host cores in
    both modes BASELINE.md section 3 names: `faithful` -- materialises X - U V^T like :337 AND normalises every active
    Laplacian anew each iteration like :349 (oracle `rebuild_laplacians`): the reference-equivalent number, reported as
    `value` -- and `lean` (Gram-identity objective, cached normalised Laplacians).

    At FULL size when the host has the memory (float64 X + two temporaries of its size + slack: BASELINE.md section 3's
    ">= 3 timed iterations after 1 warm-up at 50000 x 20000"); otherwise a bounded sample: the first m2 = min(m, 25000)
    sample rows and the first m1 = m2 / 2, all genes, the per-iteration time fitted as t(m) = a + b m through the two
    sizes, so that the m-independent part (V-side products, pathway terms, objective pieces) is NOT multiplied by
    m / m_sample.  `sample` says which."""
    from oracle import prmf_oracle as orc
    info = host_info()
    # `cores` = the thread pool that ran the GEMMs (the BLAS pool; the OpenMP pool of the same process can be larger)
    pools = info.get("blas") or []
    blas_pools = [p.get("num_threads") or 1 for p in pools if p.get("user_api") == "blas"]
    threads = max(blas_pools or [p.get("num_threads") or 1 for p in pools] or [os.cpu_count() or 1])
    n, k = V0.shape
    need_gb = 3.4 * m_full * n * 8 / 1e9 + 4.0            # X, X - U V^T, U V^T (float64) + the factors and slack
    avail = info.get("mem_available_gb")
    full = int(Xd.shape[0]) == m_full and avail is not None and avail >= need_gb
    if size == "sample":
        full = False
    elif size == "full":
        if not full:
            raise SystemExit("--cpu-baseline-size full: %.0f GB of host memory needed, %s available" % (need_gb, avail))
    m2 = m_full if full else min(int(Xd.shape[0]), 25000)
    m1 = m2 if full else (m2 // 2 if m2 < m_full else m2)
    X2 = np.empty((m2, n), dtype=np.float64)
    for r0 in range(0, m2, 8192):                         # chunked: no second full-size float64 copy on the device
        X2[r0:r0 + 8192] = Xd[r0:min(r0 + 8192, m2)].to(torch.float64).cpu().numpy()
    U2 = U0[:m2].to(torch.float64).cpu().numpy()
    Vs = V0.to(torch.float64).cpu().numpy()
    pws = [orc.pathway_from_edges(n, *[pathways[p][i] for i in (0, 2, 1)]) for p in active]

    def time_mode(lean, m_s, share, min_iters):
        X, U, V = X2[:m_s], U2[:m_s].copy(), Vs.copy()
        nsq = float(np.einsum("ij,ij->", X, X)) if lean else None
        kw = dict(n_steps=1, gamma=gamma, delta=delta, lean=lean, norm_X_sq=nsq, rebuild_laplacians=not lean)
        U, V, _ = orc.nmf_manifold_vec_update(X, U, V, pws, **kw)          # warm-up
        t0 = time.perf_counter()
        iters = 0
        while True:
            U, V, _ = orc.nmf_manifold_vec_update(X, U, V, pws, **kw)
            iters += 1
            el = time.perf_counter() - t0
            if (el > share and iters >= min_iters) or iters >= 50:
                return el / iters, iters

    out = {}
    for mode, lean, frac in (("faithful", False, 0.7), ("lean", True, 0.3)):
        t2, i2 = time_mode(lean, m2, budget_s * frac * (0.65 if m1 < m2 else 1.0), 3 if full else 2)
        if m1 < m2:
            t1, i1 = time_mode(lean, m1, budget_s * frac * 0.35, 2)
            b = (t2 - t1) / (m2 - m1)
            a = t2 - b * m2
            if a < 0.0 or b <= 0.0:           # timing noise: fall back to proportional scaling of the larger sample
                a, b = 0.0, t2 / m2
            t_full = a + b * m_full
            out[mode] = {"it_per_s": 1.0 / t_full, "s_per_iteration_full": t_full, "fit_a_s": a, "fit_b_s_per_row": b,
                         "samples": [{"rows": m1, "s_per_iteration": t1, "iterations": i1},
                                     {"rows": m2, "s_per_iteration": t2, "iterations": i2}]}
        else:
            scale = m_full / float(m2)          # 1 at full size; a rank's rows only when bench ran sharded (not used: N = 1 only)
            out[mode] = {"it_per_s": 1.0 / (t2 * scale), "s_per_iteration_full": t2 * scale,
                         "samples": [{"rows": m2, "s_per_iteration": t2, "iterations": i2}]}
    desc = ("oracle/prmf_oracle.py, float64, all %d genes, k=%d, %d active pathways; faithful mode (materialised residual "
            "as prmf_runner.py:337, every active Laplacian normalised anew per iteration as :349) is `value`, lean mode "
            "(Gram-identity objective, cached normalised Laplacians) is `lean_value`; " % (n, k, len(active)))
    if m1 < m2:
        desc += ("host memory %.0f GB available < %.0f GB needed for the full matrix: timed on the first %d and %d of %d "
                 "sample rows and extrapolated with t(m) = a + b m (m-independent cost not scaled)"
                 % (avail or -1, need_gb, m1, m2, m_full))
    else:
        desc += "timed at FULL size (%d x %d), %d + %d iterations after one warm-up each, nothing scaled" % (
            m2, n, out["faithful"]["samples"][-1]["iterations"], out["lean"]["samples"][-1]["iterations"])
    return {"value": out["faithful"]["it_per_s"], "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": desc, "full_size": bool(m1 == m2 == m_full), "lean_value": out["lean"]["it_per_s"], "modes": out, "host": info,
            # how `value` was derived, so that lines of different rounds / boxes are comparable: v3 = faithful mode rebuilds the
            # active normalised Laplacians every iteration (prmf_runner.py:349; rounds 1-3 cached them: v1 sampled, v2 two-point fit)
            "method": {"version": 3, "rebuild_laplacians": True, "materialised_residual": True,
                       "size": "full" if m1 == m2 == m_full else "two-point row sample, t(m) = a + b m", "requested": size}}


def outer_loop(torch, Xd, n, k, P, x_dtype, seed, iterations=100):
    """SURVEY 8(d)'s second measurement: the full outer loop -- nmf_pathway (prmf_runner.py:715-775): rounds of 10
    iterations with per-round pathway sampling, candidate scoring, `restrict`, the matching, best-iterate copies -- at the
    benchmark shape, wall-clock, from a float32 host matrix (so the upload is inside).  First call of the process and a
    second, warm one."""
    from prmf_amd import PathwayLibrary, solver
    from prmf_amd import synthetic as syn
    X = np.empty(tuple(Xd.shape), dtype=np.float32)
    for r0 in range(0, X.shape[0], 16384):
        X[r0:r0 + 16384] = Xd[r0:r0 + 16384].float().cpu().numpy()
    lib = PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, seed))
    marks = []
    orig_step = solver.PrmfEngine.step

    def timed_step(self, *a, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = orig_step(self, *a, **kw)
        torch.cuda.synchronize()
        marks.append(time.perf_counter() - t0)
        return r

    calls = []
    solver.PrmfEngine.step = timed_step
    try:
        # first call of the process, a warm one (both from the float32 HOST matrix: the upload is inside), and one with the
        # matrix already on the GPU in the storage type (what the CLI's on-device normalisation hands over)
        for src in (X, X, Xd):
            np.random.seed(seed)
            marks.clear()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                solver.nmf_pathway(src, None, library=lib, k_latent=k, max_iter=iterations, x_dtype=x_dtype, tol=0.0)
            calls.append((time.perf_counter() - t0, sum(marks), len(marks)))
    finally:
        solver.PrmfEngine.step = orig_step
    (t_first, in_first, rounds), (t_warm, in_warm, _), (t_dev, in_dev, _) = calls
    return {"what": "nmf_pathway (outer solver, prmf_runner.py:556-792): %d iterations = %d rounds of 10 with pathway sampling, "
                    "candidate scoring / restrict / matching, best-iterate copies; float32 host X uploaded inside "
                    "(`*_device_resident`: X already on the GPU in the storage type)" % (iterations, rounds),
            "iterations": iterations, "rounds": rounds,
            "seconds_first_call": t_first, "seconds_warm_call": t_warm, "seconds_inner_iterations": in_warm,
            "it_per_s_warm": iterations / t_warm, "it_per_s_first_call": iterations / t_first,
            "it_per_s_inner_only": iterations / in_warm,
            "seconds_device_resident": t_dev, "it_per_s_device_resident": iterations / t_dev,
            "host_matrix_gb": X.nbytes / 1e9}


def run_rank(args, ranks, proc):
    """One rank of the benchmark.  `ranks`: prmf_amd.ranks protocol (processes or threads); proc: the process-mode
    settings (local rank, backend) or None for thread ranks."""
    import torch
    from prmf_amd import PathwayLibrary
    from prmf_amd import synthetic as syn
    from prmf_amd.solver import PrmfEngine

    world, rank = ranks.world, ranks.rank
    multi = ranks.multi          # several ranks -- or ONE rank driven through the multi-rank code (PRMF_FORCE_RANK_PATH, prmf_amd/ranks.py)
    dev_index = proc["dev_index"] if proc else (ranks.device_index or 0)
    dev = "cuda:%d" % dev_index
    m, n, k, P, x_dtype = syn.CONFIGS[args.workload]
    tdt = torch.bfloat16 if x_dtype == "bf16" else torch.float32
    lo, hi = syn.shard_rows(m, world, rank)
    Xd = syn.x_rows_torch(torch, dev, args.seed, n, k, lo, hi, tdt)
    eng = PrmfEngine(hi - lo, n, k, x_dtype=x_dtype, device=dev_index, nranks=world, rank=rank,
                     dist_mode=args.dist_mode if multi else None, eager_objective=args.eager_objective)
    nsq = ranks.sum_float(eng.set_x_local(Xd), dev)           # one pass: padded copy, padded transpose, ||X||^2 of this rank's rows
    norm_x = math.sqrt(nsq)
    eng.set_norm_x_sq(nsq)
    pathways = syn.tree_pathways(n, P, 300, args.seed)
    eng.set_library(PathwayLibrary.from_edge_lists(n, pathways))
    active = [kk % P for kk in range(k)]
    eng.set_active(active)
    gamma, delta = norm_x / k, 10.0 / norm_x                     # CLI defaults (prmf_runner.py:644-645)
    U = syn.init_factor_torch(torch, dev, args.seed + 1, m, k, lo, hi)
    V = syn.init_factor_torch(torch, dev, args.seed + 2, n, k)
    U0, V0 = U.clone(), V.clone()

    runner = None
    if multi:
        from prmf_amd.distributed import DistributedPrmf
        # The driver-visible line never reports a slow path silently: the in-library RCCL step, or an error on every rank.
        # (rehearsal with the RCCL test double, PRMF_RCCL_LIB: the in-library step although the backend is gloo;
        #  PRMF_BENCH_BACKEND=gloo without it: host-driven, declared in the line)
        host_driven = proc is not None and proc["backend"] != "nccl" and not os.environ.get("PRMF_RCCL_LIB")
        runner = DistributedPrmf(eng, group=None if proc else ranks, native=not host_driven)
        step = lambda s: runner.step(U, V, s, gamma, delta)
    else:
        step = lambda s: eng.step(U, V, s, gamma, delta)

    def fence():
        ranks.barrier()
        torch.cuda.synchronize(dev_index)

    if args.warmup > 0:
        step(args.warmup)
    # The timed block is EXACTLY --steps iterations between two fences (barrier + device synchronise), max over
    # ranks.  It is repeated --reps times back to back (every repetition continues the same solve: real work, same
    # cost); `ms_per_step` / `value` are those of the MEDIAN repetition, the spread is reported next to them.
    rep_s = []
    for _ in range(max(1, args.reps)):
        fence()
        t0 = time.perf_counter()
        objs, _ = step(args.steps)
        fence()
        rep_s.append(time.perf_counter() - t0)
    rep_s = sorted(ranks.max_floats(rep_s, dev))
    elapsed = rep_s[len(rep_s) // 2] if len(rep_s) % 2 else 0.5 * (rep_s[len(rep_s) // 2 - 1] + rep_s[len(rep_s) // 2])

    # live roofline of the dominant kernel (the X / X^T sweep): HIP events on the
    # launch stream around every sweep launch of a short extra run
    eng.set_timing(True)
    step(min(args.steps, 20))
    torch.cuda.synchronize(dev_index)
    ms_xv, ms_xtu, _ = eng.timing() if hi > lo else (float("nan"), float("nan"), 0)
    eng.set_timing(False)
    eb = 2 if x_dtype == "bf16" else 4
    ml = hi - lo
    # algorithmic bytes per sweep launch: the data matrix once + the factor once + the product once
    bytes_launch = 0.5 * ((ml * n * eb + n * k * 4 + ml * k * 4) + (ml * n * eb + ml * k * 4 + n * k * 4))
    avg_ms = 0.5 * (ms_xv + ms_xtu)
    achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
    # Every sweep runs on the bf16 MFMA (bf16 X: 2 instructions per 16-element product with the hi/lo factor; f32 X: 5, the
    # three-piece split).  At the nominal 2.5 PFLOP/s even the heaviest case (f32 X, k = 128: 5 x 2 m n k = 8e11 flop
    # per launch = 0.32 ms) is no slower than its HBM time at 8 TB/s (0.31 ms), so HBM is the roof for every workload;
    # DESIGN.md section 5 has the clock the chip actually sustains under these kernels.
    flops_launch = 2.0 * ml * n * k
    tflops = flops_launch / (avg_ms * 1e-3) / 1e12
    # whole-iteration roofline: B = 2 m n s_X + 12 (m + n) k + 8 sum nnz + 12 sum |S|
    nnz_active = sum(2 * len(pathways[p][0]) for p in active)
    sup_active = sum(len(pathways[p][2]) for p in active)
    bytes_iter = 2.0 * ml * n * eb + 12.0 * (ml + n) * k + 8.0 * nnz_active + 12.0 * sup_active

    extras = {}
    if not multi and not args.no_extras:
        # (1) calibration: a pure non-temporal stream of the resident X and X^T on THIS box, right next to the timing pass
        sx_ms, sx_b = eng.stream_read("x", 10)
        st_ms, st_b = eng.stream_read("xt", 10)
        stream_gbs = 0.5 * (sx_b / sx_ms + st_b / st_ms) / 1e6
        # the sweep streams the same padded bytes (+ factor and slabs): its rate on those bytes against the pure stream's
        sweep_padded_gbs = 0.5 * (sx_b / ms_xv + st_b / ms_xtu) / 1e6
        extras["stream"] = {"stream_gbs": stream_gbs, "stream_ms_x": sx_ms, "stream_ms_xt": st_ms, "bytes_x": sx_b, "bytes_xt": st_b,
                            "frac_of_stream": sweep_padded_gbs / stream_gbs, "stream_frac_of_peak": stream_gbs / HBM_PEAK_GBS}
        # (2) update-only rate: the iterations without the evaluation of the objective records (the reference cannot
        # separate them) -- phase calls between prmf_begin and a device synchronise; prmf_end (the two record
        # launches + the copy of the records) comes after the clock has stopped
        if not args.eager_objective and args.steps <= 256:
            ups = []
            for _ in range(3):
                eng.begin(U, V, gamma, delta)
                torch.cuda.synchronize(dev_index)
                t0 = time.perf_counter()
                for _i in range(args.steps):
                    eng.phase_u()
                    eng.phase_q()
                    eng.phase_v()
                    eng.phase_post()
                torch.cuda.synchronize(dev_index)
                ups.append(time.perf_counter() - t0)
                eng.end(args.steps)
            up = sorted(ups)[1]
            extras["update_only"] = {"value": args.steps / up, "unit": "iterations/s", "ms_per_step": 1e3 * up / args.steps,
                                     "what": "U-update + V-update incl. what the records need kept per iteration; "
                                             "records evaluated after the clock stops"}

    out = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.workload)
        if not multi and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_sweep_launch")
        native = runner.native if runner else None
        out = {
            "metric": "nmf_update_iterations_per_sec", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if x_dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic X %dx%d (%s), k=%d, %d random tree pathways of 300 genes, "
                                   "active pathway k mod P, gamma=||X||/k, delta=10/||X||"
                                   % (args.workload, m, n, x_dtype, k, P),
                       "samples": m, "genes": n, "k_latent": k, "pathways": P,
                       # 2 sweeps + 2 fused epilogues at every k (k > 64: the column-split epilogue); per-iteration objective: + 3
                       "launches_per_iteration": None if multi else (7 if args.eager_objective else 4),
                       "parallelism": ("single GPU" if not multi else
                                       "X row-sharded x%d, V update %s, RCCL %s" % (
                                           world, eng.dist_mode, "in-library" if native else "host-driven"))},
            "recon_err": objs[-1]["recon"], "objective": objs[-1]["obj"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "tflops": tflops},
        }
        out["roofline"].update({"traffic": traffic,
                                "kernel": "sweep_kernel (X*V and X^T*U, 2 launches/iteration)",
                                "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_flops_per_launch": flops_launch,
                                "avg_launch_ms": avg_ms, "ms_xv": ms_xv, "ms_xtu": ms_xtu,
                                "sweep_share_of_step": 2 * avg_ms / (1e3 * elapsed / args.steps),
                                "iteration_bytes": bytes_iter,
                                "iteration_frac": bytes_iter / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS})
        if "stream" in extras:
            out["roofline"].update(extras["stream"])
        if "update_only" in extras:
            out["update_only"] = extras["update_only"]
        out["reps"] = {"n": len(rep_s), "ms_per_step_median": 1e3 * elapsed / args.steps,
                       "ms_per_step_min": 1e3 * rep_s[0] / args.steps, "ms_per_step_max": 1e3 * rep_s[-1] / args.steps}
        if multi:
            ci = eng.comm_info()
            out["distributed"] = {
                "mode": ci["mode"], "collectives": "in-library RCCL (prmf_step_dist)" if native else "host-driven (torch.distributed)",
                "ranks_are": "processes (torch.distributed.run)" if proc else "threads of one process (prmf_amd.ranks.LocalRanks)",
                "rccl_comm_world_size": ci["comm_nranks"], "world_size": world,
                "exchanges_per_iteration": (["one RCCL group: all_reduce U^T U (k*k f64) + reduce_scatter X^T U (n*k/N f32 per rank)",
                                             "all_gather V (n*k/N f32 per rank + the own-column update's sums)"] if ci["mode"] == "colshard" else
                                            ["one RCCL group: all_reduce U^T U (k*k f64) + all_reduce X^T U (n*k f32)"]),
                "phase_ms_rank0": dict(zip(["phase_u", "phase_q+exchange", "phase_v+exchange", "phase_post"], ci["ms_phase"]))}
        if os.environ.get("PRMF_RCCL_LIB") or (proc and proc["backend"] != "nccl"):
            out["config"]["rehearsal"] = ("ranks share GPU(s), exchanges through %s: the sequence is the real one, the time is not a measurement"
                                          % ("the RCCL test double" if os.environ.get("PRMF_RCCL_LIB") else proc["backend"]))
    eng.close()
    if rank == 0 and not multi:
        if not args.no_extras and args.workload in ("c2", "c3", "c3f32", "c5shard", "c5shardf32"):
            out["outer_loop"] = outer_loop(torch, Xd, n, k, P, x_dtype, args.seed)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, Xd, U0, V0, pathways, active, gamma, delta, m, args.cpu_seconds, args.cpu_baseline_size)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PRMF_FORCE_RANK_PATH=1 under the launcher with ONE process: the branch below at world size 1 -- NCCL process group, rank
    # protocol over it, RCCL communicator bootstrap, prmf_step_dist -- i.e. what the 8-GPU launch executes, on a one-GPU box
    forced = bool(os.environ.get("PRMF_FORCE_RANK_PATH")) and "RANK" in os.environ
    if world > 1 or forced:                        # one process per GPU, started by torch.distributed.run
        import torch
        import torch.distributed as dist
        from prmf_amd.ranks import TorchRanks
        args.gpus = world
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # rehearsal on a one-GPU box: PRMF_BENCH_BACKEND=gloo PRMF_BENCH_SHARE_GPU=1 lets N ranks share cuda:0 and
        # exchange through the host (RCCL refuses two ranks on one device); never used for reported numbers
        backend = os.environ.get("PRMF_BENCH_BACKEND", "nccl")
        if os.environ.get("PRMF_BENCH_SHARE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:%d" % local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        try:
            out = run_rank(args, TorchRanks(None), {"dev_index": local_rank, "backend": backend})
        except BaseException:
            # no barrier on the way out of a failure (the peers may sit in another collective): exit non-zero, the launcher ends the job
            raise
        if out is not None:
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
    elif args.gpus > 1:                            # N thread ranks inside this process
        from prmf_amd.ranks import LocalRanks
        sys.stderr.write("bench.py: --gpus %d without a launcher (no RANK / WORLD_SIZE in the environment): running %d THREAD ranks in "
                         "this process, rank r on device r %% device_count; for one process per GPU start it under "
                         "`python -m torch.distributed.run --nproc-per-node %d`\n" % (args.gpus, args.gpus, args.gpus))
        outs = LocalRanks(args.gpus).run(lambda view: run_rank(args, view, None))
        print(json.dumps(outs[0]), flush=True)
    else:
        import torch
        from prmf_amd.ranks import TorchRanks
        torch.cuda.set_device(0)
        print(json.dumps(run_rank(args, TorchRanks(None), {"dev_index": 0, "backend": "nccl"})), flush=True)


if __name__ == "__main__":
    main()
