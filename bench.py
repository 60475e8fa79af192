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
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak at the nominal clock (same guide: ~2.5 PF dense, never the 2:1-sparsity figure)
NOMINAL_CLOCK_GHZ = 2.4


def stored_counters(workload):
    """What the counter passes of this round left under profiles/ for `workload`, IF they were taken on the sweep kernel this
    tree builds (`sweep_sources_sha16`, prmf_amd/_build.py): HBM bytes per sweep launch (separate --pmc FETCH_SIZE / WRITE_SIZE
    passes, FETCH x2 on gfx950: tools/pmc_summary.py) and the shader clock held under the kernel (GRBM_GUI_ACTIVE / 8 /
    duration: tools/pmc_mfma_summary.py).  A measurement of another kernel is not reported."""
    from prmf_amd._build import sweep_sources_sha16
    stamp = sweep_sources_sha16()
    out = {"traffic": None, "clock_ghz": None, "sweep_sources_sha16": stamp, "stale": []}
    for key, fname, field in (("traffic", "pmc_traffic_%s.json", "hbm_bytes_per_sweep_launch"), ("clock_ghz", "pmc_mfma_%s.json", "effective_clock_ghz")):
        path = os.path.join(ROOT, "profiles", fname % workload)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        if d.get("sweep_sources_sha16") == stamp:
            out[key] = d.get(field)
        else:
            out["stale"].append(os.path.basename(path))
    return out


def sweep_roofline(workload, x_dtype, ml, n, k, ms_xv, ms_xtu, step_s, nnz_active, sup_active):
    """The roofline block of one workload from the average durations of its two sweep launches (HIP events on the launch
    stream, this run) and the duration of a whole iteration.  BOTH bounds are priced and `bound` names the one that binds:
      hbm   algorithmic bytes per launch (the data matrix once + the factor once + the product once: SURVEY 8d) / 8 TB/s
      mfma  matrix instructions ISSUED per launch -- every product runs hi + lo of the factor (2 bf16 MFMAs; f32 X: three
            exact pieces of X against hi / lo, 5) on the latent columns padded to 32 -- / the dense bf16 peak AT THE CLOCK THE
            CHIP HOLDS under this kernel (counter measurement, profiles/pmc_mfma_<workload>.json; nominal 2.4 GHz if none)."""
    eb = 2 if x_dtype == "bf16" else 4
    kp = (k + 31) // 32 * 32
    bytes_launch = 0.5 * ((ml * n * eb + n * k * 4 + ml * k * 4) + (ml * n * eb + ml * k * 4 + n * k * 4))
    avg_ms = 0.5 * (ms_xv + ms_xtu)
    gbs = bytes_launch / (avg_ms * 1e-3) / 1e9
    flops_launch = 2.0 * ml * n * k                                   # the product itself
    issued = (2 if x_dtype == "bf16" else 5) * 2.0 * ml * n * kp      # what the matrix pipe executes for it
    tflops_issued = issued / (avg_ms * 1e-3) / 1e12
    sc = stored_counters(workload)
    clock = sc["clock_ghz"]
    if clock and clock > NOMINAL_CLOCK_GHZ:        # short dispatches read high (GRBM_GUI_ACTIVE over a 40 us launch): never above the spec
        clock = NOMINAL_CLOCK_GHZ
    peak_at_clock = MFMA_BF16_PEAK_TFLOPS * (clock / NOMINAL_CLOCK_GHZ) if clock else None
    t_hbm = bytes_launch / (HBM_PEAK_GBS * 1e9)
    t_mfma = issued / ((peak_at_clock or MFMA_BF16_PEAK_TFLOPS) * 1e12)
    bound = "mfma" if t_mfma > t_hbm else "hbm"
    bytes_iter = 2.0 * ml * n * eb + 12.0 * (ml + n) * k + 8.0 * nnz_active + 12.0 * sup_active
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "floor_ms": 1e3 * t_hbm}
    mfma = {"achieved": tflops_issued, "peak": peak_at_clock or MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops_issued / (peak_at_clock or MFMA_BF16_PEAK_TFLOPS), "floor_ms": 1e3 * t_mfma,
            "peak_nominal": MFMA_BF16_PEAK_TFLOPS, "frac_of_nominal": tflops_issued / MFMA_BF16_PEAK_TFLOPS,
            "clock_ghz": clock, "clock_source": ("profiles/pmc_mfma_%s.json" % workload) if clock else "none for this kernel: nominal %.1f GHz" % NOMINAL_CLOCK_GHZ,
            "issued_flops_per_launch": issued, "mfma_per_product": 2 if x_dtype == "bf16" else 5, "k_padded": kp}
    top = hbm if bound == "hbm" else mfma
    rf = {"bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
          "traffic": sc["traffic"], "traffic_ratio": (sc["traffic"] / bytes_launch) if sc["traffic"] else None,
          "traffic_source": {"sweep_sources_sha16": sc["sweep_sources_sha16"], "stale_measurements_not_reported": sc["stale"]},
          "hbm": hbm, "mfma": mfma, "tflops": flops_launch / (avg_ms * 1e-3) / 1e12,
          "kernel": "sweep_kernel (X*V and X^T*U, 2 launches/iteration)",
          "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_flops_per_launch": flops_launch,
          "avg_launch_ms": avg_ms, "ms_xv": ms_xv, "ms_xtu": ms_xtu,
          "sweep_share_of_step": 2 * avg_ms / (1e3 * step_s),
          "iteration_bytes": bytes_iter, "iteration_frac": bytes_iter / step_s / 1e9 / HBM_PEAK_GBS,
          "iteration_floor_ms": 2e3 * max(t_hbm, t_mfma), "iteration_frac_of_binding_floor": 2 * max(t_hbm, t_mfma) / step_s}
    return rf


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
    ap.add_argument("--secondary", default="auto",
                    help="the other BASELINE workloads measured after the headline and reported under `secondary`: a comma list of "
                         "c2, c3f32, c5, c4_share_of_2, c4_share_of_4, c4_share_of_8, c5_share_of_8; `auto` (default): all of them after the c3 headline, none "
                         "after another workload; `none`")
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
    benchmark shape, wall-clock, from a float32 host matrix (so the upload is inside).  First call of the process, then
    three warm ones (median reported, all three listed), then three with the matrix already on the GPU."""
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
        # (three warm calls of each kind, the median reported: one call is 0.15-0.25 s of which a third is host work, and a
        #  single descheduled moment of a shared host showed up as 0.25 s where the next call took 0.16)
        for src in (X, X, X, X, Xd, Xd, Xd):
            np.random.seed(seed)
            marks.clear()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                solver.nmf_pathway(src, None, library=lib, k_latent=k, max_iter=iterations, x_dtype=x_dtype, tol=0.0)
            calls.append((time.perf_counter() - t0, sum(marks), len(marks)))
    finally:
        solver.PrmfEngine.step = orig_step
    (t_first, in_first, rounds) = calls[0]
    (t_warm, in_warm, _) = sorted(calls[1:4])[1]
    (t_dev, in_dev, _) = sorted(calls[4:7])[1]
    warm_all, dev_all = [c[0] for c in calls[1:4]], [c[0] for c in calls[4:7]]
    return {"what": "nmf_pathway (outer solver, prmf_runner.py:556-792): %d iterations = %d rounds of 10 with pathway sampling, "
                    "candidate scoring / restrict / matching, best-iterate copies; float32 host X uploaded inside "
                    "(`*_device_resident`: X already on the GPU in the storage type)" % (iterations, rounds),
            "iterations": iterations, "rounds": rounds,
            "seconds_first_call": t_first, "seconds_warm_call": t_warm, "seconds_inner_iterations": in_warm,
            "it_per_s_warm": iterations / t_warm, "it_per_s_first_call": iterations / t_first,
            "it_per_s_inner_only": iterations / in_warm,
            "seconds_device_resident": t_dev, "it_per_s_device_resident": iterations / t_dev,
            "seconds_warm_calls": warm_all, "seconds_device_resident_calls": dev_all,
            "host_matrix_gb": X.nbytes / 1e9}


def measure_single(torch, name, seed, steps, warmup, reps):
    """One more workload on this GPU, measured like the headline (median of `reps` timed blocks of `steps` iterations, then a
    short pass with HIP events around the sweep launches) -- the numbers the driver's record had no line for until round 6."""
    from prmf_amd import PathwayLibrary
    from prmf_amd import synthetic as syn
    from prmf_amd.solver import PrmfEngine
    m, n, k, P, x_dtype = syn.CONFIGS[name]
    dev = "cuda:%d" % torch.cuda.current_device()
    Xd = syn.x_rows_torch(torch, dev, seed, n, k, 0, m, torch.bfloat16 if x_dtype == "bf16" else torch.float32)
    eng = PrmfEngine(m, n, k, x_dtype=x_dtype)
    try:
        norm_x = math.sqrt(eng.set_x(Xd))
        del Xd
        pathways = syn.tree_pathways(n, P, 300, seed)
        eng.set_library(PathwayLibrary.from_edge_lists(n, pathways))
        active = [kk % P for kk in range(k)]
        eng.set_active(active)
        gamma, delta = norm_x / k, 10.0 / norm_x
        U = syn.init_factor_torch(torch, dev, seed + 1, m, k)
        V = syn.init_factor_torch(torch, dev, seed + 2, n, k)
        eng.step(U, V, warmup, gamma, delta)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            objs, _ = eng.step(U, V, steps, gamma, delta)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        el = sorted(ts)[len(ts) // 2]
        eng.set_timing(True)
        eng.step(U, V, min(steps, 20), gamma, delta)
        torch.cuda.synchronize()
        ms_xv, ms_xtu, _ = eng.timing()
        rf = sweep_roofline(name, x_dtype, m, n, k, ms_xv, ms_xtu, el / steps,
                            sum(2 * len(pathways[p][0]) for p in active), sum(len(pathways[p][2]) for p in active))
        return {"workload": "%dx%d (%s), k=%d, P=%d" % (m, n, x_dtype, k, P), "it_per_s": steps / el, "ms_per_step": 1e3 * el / steps,
                "steps": steps, "reps": reps, "sweep_ms": [ms_xv, ms_xtu], "bound": rf["bound"], "frac": rf["frac"],
                "hbm_frac": rf["hbm"]["frac"], "mfma_frac": rf["mfma"]["frac"], "mfma_clock_ghz": rf["mfma"]["clock_ghz"],
                "iteration_frac": rf["iteration_frac"], "iteration_frac_of_binding_floor": rf["iteration_frac_of_binding_floor"],
                "traffic_ratio": rf["traffic_ratio"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                "issued_flops_per_launch": rf["mfma"]["issued_flops_per_launch"], "recon_err": objs[-1]["recon"]}
    finally:
        eng.close()


def measure_share(torch, name, nranks, seed, iters, whole_ms):
    """Device time of ONE rank's share of an N-rank run on this GPU: the real kernel sequence of the column-sharded mode at
    the real share shape through the host-driven phase calls, with device copies standing in for the two exchanges (no
    wire: the numbers of the solve are meaningless, the kernels and their sizes are the real ones).  `whole_ms`: the
    one-GPU iteration of the global problem measured in this run -> the scaling the share allows BEFORE any exchange."""
    from prmf_amd import PathwayLibrary
    from prmf_amd import synthetic as syn
    from prmf_amd.solver import PrmfEngine
    m, n, k, P, x_dtype = syn.CONFIGS[name]
    dev = "cuda:%d" % torch.cuda.current_device()
    lo, hi = syn.shard_rows(m, nranks, 0)
    Xd = syn.x_rows_torch(torch, dev, seed, n, k, lo, hi, torch.bfloat16 if x_dtype == "bf16" else torch.float32)
    eng = PrmfEngine(hi - lo, n, k, x_dtype=x_dtype, nranks=nranks, rank=0, dist_mode="colshard")
    try:
        eng.set_x(Xd, norm_x_sq=float(nranks) * eng.local_norm_sq(Xd))
        del Xd
        eng.set_library(PathwayLibrary.from_edge_lists(n, syn.tree_pathways(n, P, 300, seed)))
        eng.set_active([kk % P for kk in range(k)])
        U = syn.init_factor_torch(torch, dev, seed + 1, m, k, lo, hi)
        V = syn.init_factor_torch(torch, dev, seed + 2, n, k)
        b = eng.comm_buffers()
        eng.set_timing(True)
        ts = []
        for _ in range(3):
            V.copy_(syn.init_factor_torch(torch, dev, seed + 2, n, k))
            eng.invalidate_factors()
            eng.begin(U, V, 1.0, 1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _i in range(iters):
                eng.phase_u()
                eng.phase_q()
                b["q_recv"].copy_(b["q_send"][:b["q_recv"].numel()])                              # stand-in for the reduce-scatter
                eng.phase_v()
                b["v_recv"].view(nranks, -1).copy_(b["v_send"].unsqueeze(0).expand(nranks, -1))   # stand-in for the all-gather
                eng.phase_post()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            try:
                eng.end(iters)
            except FloatingPointError:
                pass
        ms = 1e3 * sorted(ts)[1] / iters
        ms_xv, ms_xtu, _ = eng.timing()
        # the two stand-in copies alone (they are not part of a real run -- the exchanges are, whose time no single GPU can measure):
        # subtracted from the kernels' figure below
        tc = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _i in range(iters):
                b["q_recv"].copy_(b["q_send"][:b["q_recv"].numel()])
                b["v_recv"].view(nranks, -1).copy_(b["v_send"].unsqueeze(0).expand(nranks, -1))
            torch.cuda.synchronize()
            tc.append(time.perf_counter() - t0)
        ms_copies = 1e3 * sorted(tc)[1] / iters
        ms_kernels = ms - ms_copies
        eb = 2 if x_dtype == "bf16" else 4
        ml = hi - lo
        bytes_launch = ml * n * eb + 2.0 * (n + ml) * k           # average of the two sweeps: matrix once + factor once + product once
        bytes_iter = 2.0 * ml * n * eb + 12.0 * (ml + n) * k
        return {"workload": "rank 0 of %d of %dx%d (%s), k=%d: %d rows" % (nranks, m, n, x_dtype, k, ml), "ms_per_iteration_device": ms,
                "stand_in_copies_ms": ms_copies, "ms_per_iteration_kernels": ms_kernels,
                "iterations": iters, "sweep_ms": [ms_xv, ms_xtu], "other_kernels_ms": ms - ms_xv - ms_xtu,
                "sweep_hbm_frac": bytes_launch / (0.5e-3 * (ms_xv + ms_xtu)) / 1e9 / HBM_PEAK_GBS,
                "iteration_frac": bytes_iter / (ms_kernels * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "whole_problem_ms_one_gpu": whole_ms, "speedup_bound_before_exchanges": (whole_ms / ms_kernels) if whole_ms else None,
                "note": "ms_per_iteration_device includes two device copies that stand in for the reduce-scatter and the all-gather "
                        "(stand_in_copies_ms, timed alone); ms_per_iteration_kernels, iteration_frac and the speed-up bound are without them. "
                        "No wire time is in any of these numbers.  The host-driven phase calls measured here launch the U^T U reduction "
                        "on its own (~5 us); the in-library step (prmf_step_dist) runs it inside the X^T U sweep"}
    finally:
        eng.close()


def secondary_workloads(torch, args, c3_ms, names):
    """The workloads BASELINE.json names besides the headline, each measured in this run on this box."""
    out = {}
    plan = (("c2", 40, 10, 3), ("c3f32", 20, 5, 3), ("c5", 10, 3, 3))
    for name, steps, warmup, reps in plan:
        if name not in names:
            continue
        try:
            out[name] = measure_single(torch, name, args.seed, steps, warmup, reps)
        except Exception as e:        # a workload that does not fit this box must not take the headline line with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    # one rank's share of the N-rank partitions of C3 (N = 2, 4, 8 = C4) and of C5 (N = 8): what a single GPU can say about the
    # scaling curve -- the kernels of a rank at the real share shape, before any exchange
    for name, whole, label, nr in (("c3", c3_ms, "c4_share_of_2", 2), ("c3", c3_ms, "c4_share_of_4", 4), ("c3", c3_ms, "c4_share_of_8", 8),
                                   ("c5", out.get("c5", {}).get("ms_per_step"), "c5_share_of_8", 8)):
        if label not in names:
            continue
        try:
            out[label] = measure_share(torch, name, nr, args.seed, 30, whole)
        except Exception as e:
            out[label] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    if c3_ms:
        curve = {str(nr): out[lb]["speedup_bound_before_exchanges"] for lb, nr in (("c4_share_of_2", 2), ("c4_share_of_4", 4), ("c4_share_of_8", 8))
                 if lb in out and "error" not in out[lb]}
        if curve:
            out["c3_scaling_bound_before_exchanges"] = dict({"1": 1.0}, **curve)
    return out


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
    ml = hi - lo
    nnz_active = sum(2 * len(pathways[p][0]) for p in active)
    sup_active = sum(len(pathways[p][2]) for p in active)
    rf = sweep_roofline(args.workload, x_dtype, ml, n, k, ms_xv, ms_xtu, elapsed / args.steps, nnz_active, sup_active)

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
            "roofline": rf,
        }
        if multi:
            rf["traffic"] = rf["traffic_ratio"] = None          # the counter passes were taken on the single-GPU shapes
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
        names = {"auto": ["c2", "c3f32", "c5", "c4_share_of_2", "c4_share_of_4", "c4_share_of_8", "c5_share_of_8"] if (args.workload == "c3" and not args.no_extras) else [],
                 "none": []}.get(args.secondary, [x for x in args.secondary.split(",") if x])
        if names:
            del Xd, U, V, U0, V0
            torch.cuda.empty_cache()
            out["secondary"] = secondary_workloads(torch, args, out["ms_per_step"] if args.workload == "c3" else None, names)
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
