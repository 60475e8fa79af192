#!/usr/bin/env python
"""Benchmark of the PRMF inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c5]

One "step" = one iteration of the hot path (U-update + V-update + the objective
record, script/prmf_runner.py:420-447) on synthetic inputs of the configuration
BASELINE.json's target is quoted on (c3: X 50000 x 20000 bf16, k = 64, 300 tree
pathways).  X, U, V are resident in HBM before the timed
region.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the
SAME global problem is row-sharded over the ranks (strong scaling); per iteration the
k x k Gram U^T U is all-reduced together with the reduce-scatter of the n x k product X^T U by owner column block (one RCCL
group), the V update is sharded by latent column and V is all-gathered (--dist-mode colshard, the default everywhere), or
X^T U is all-reduced and every rank updates all of V (--dist-mode replicated); RCCL issued inside the library.  Rank 0
prints one JSON line.
"""
import argparse
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
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
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
    return info


def cpu_baseline(torch, Xd, U0, V0, pathways, active, gamma, delta, m_full, budget_s):
    """The oracle (float64 NumPy/SciPy restatement of prmf_runner.py:419-447) timed on this box's host cores in
    both modes BASELINE.md section 3 names: `faithful` (materialises X - U V^T like :337 -- the reference-equivalent
    number, reported as `value`; the oracle caches the normalised Laplacians the reference rebuilds per iteration
    at :349, so it is if anything faster than the reference) and `lean` (Gram-identity objective).  Bounded sample: the first m2 = min(m, 25000) sample rows and the
    first m1 = m2 / 2, all genes; the per-iteration time is fitted as t(m) = a + b m through the two sizes, so the
    m-independent part (V-side products, pathway terms, objective pieces) is NOT multiplied by m / m_sample; at
    m <= 25000 the full problem is timed and nothing is scaled."""
    from oracle import prmf_oracle as orc
    info = host_info()
    # `cores` = the thread pool that ran the GEMMs (the BLAS pool; the OpenMP pool of the same process can be larger)
    pools = info.get("blas") or []
    blas_pools = [p.get("num_threads") or 1 for p in pools if p.get("user_api") == "blas"]
    threads = max(blas_pools or [p.get("num_threads") or 1 for p in pools] or [os.cpu_count() or 1])
    n, k = V0.shape
    m2 = min(int(Xd.shape[0]), 25000)
    m1 = m2 // 2 if m2 < m_full else m2
    X2 = Xd[:m2].to(torch.float64).cpu().numpy()
    U2 = U0[:m2].to(torch.float64).cpu().numpy()
    Vs = V0.to(torch.float64).cpu().numpy()
    pws = [orc.pathway_from_edges(n, *[pathways[p][i] for i in (0, 2, 1)]) for p in active]

    def time_mode(lean, m_s, share):
        X, U, V = X2[:m_s], U2[:m_s].copy(), Vs.copy()
        nsq = float(np.einsum("ij,ij->", X, X)) if lean else None
        U, V, _ = orc.nmf_manifold_vec_update(X, U, V, pws, n_steps=1, gamma=gamma, delta=delta, lean=lean, norm_X_sq=nsq)
        t0 = time.perf_counter()
        iters = 0
        while True:
            U, V, _ = orc.nmf_manifold_vec_update(X, U, V, pws, n_steps=1, gamma=gamma, delta=delta, lean=lean, norm_X_sq=nsq)
            iters += 1
            el = time.perf_counter() - t0
            if (el > share and iters >= 2) or iters >= 50:
                return el / iters, iters

    out = {}
    for mode, lean, frac in (("faithful", False, 0.7), ("lean", True, 0.3)):
        t2, i2 = time_mode(lean, m2, budget_s * frac * (0.65 if m1 < m2 else 1.0))
        if m1 < m2:
            t1, i1 = time_mode(lean, m1, budget_s * frac * 0.35)
            b = (t2 - t1) / (m2 - m1)
            a = t2 - b * m2
            if a < 0.0 or b <= 0.0:           # timing noise: fall back to proportional scaling of the larger sample
                a, b = 0.0, t2 / m2
            t_full = a + b * m_full
            out[mode] = {"it_per_s": 1.0 / t_full, "s_per_iteration_full": t_full, "fit_a_s": a, "fit_b_s_per_row": b,
                         "samples": [{"rows": m1, "s_per_iteration": t1, "iterations": i1},
                                     {"rows": m2, "s_per_iteration": t2, "iterations": i2}]}
        else:
            out[mode] = {"it_per_s": 1.0 / t2, "s_per_iteration_full": t2,
                         "samples": [{"rows": m2, "s_per_iteration": t2, "iterations": i2}]}
    desc = ("oracle/prmf_oracle.py, float64, all %d genes, k=%d, %d active pathways; faithful mode (materialised residual "
            "as prmf_runner.py:337; normalised Laplacians cached, which the reference rebuilds per iteration) is `value`, lean mode (Gram-identity objective) is "
            "`lean_value`; " % (n, k, len(active)))
    if m1 < m2:
        desc += ("timed on the first %d and %d of %d sample rows and extrapolated with t(m) = a + b m "
                 "(m-independent cost not scaled)" % (m1, m2, m_full))
    else:
        desc += "timed at full size (%d rows), nothing scaled" % m2
    return {"value": out["faithful"]["it_per_s"], "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": desc, "lean_value": out["lean"]["it_per_s"], "modes": out, "host": info}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from prmf_amd import PathwayLibrary
    from prmf_amd import synthetic as syn
    from prmf_amd.solver import PrmfEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    # rehearsal on a one-GPU box: PRMF_BENCH_BACKEND=gloo PRMF_BENCH_SHARE_GPU=1 lets N ranks share cuda:0 and
    # exchange through the host (RCCL refuses two ranks on one device); never used for reported numbers
    backend = os.environ.get("PRMF_BENCH_BACKEND", "nccl")
    if os.environ.get("PRMF_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    m, n, k, P, x_dtype = syn.CONFIGS[args.workload]
    tdt = torch.bfloat16 if x_dtype == "bf16" else torch.float32
    lo, hi = syn.shard_rows(m, world, rank)
    Xd = syn.x_rows_torch(torch, dev, args.seed, n, k, lo, hi, tdt)
    eng = PrmfEngine(hi - lo, n, k, x_dtype=x_dtype, device=local_rank, nranks=world, rank=rank,
                     dist_mode=args.dist_mode if world > 1 else None, eager_objective=args.eager_objective)
    cdev = dev if backend == "nccl" else "cpu"         # where the tiny control tensors of the collectives live
    nsq = torch.tensor([eng.local_norm_sq(Xd)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(nsq)
    norm_x = math.sqrt(float(nsq.item()))
    eng.set_x(Xd, norm_x_sq=float(nsq.item()))
    pathways = syn.tree_pathways(n, P, 300, args.seed)
    eng.set_library(PathwayLibrary.from_edge_lists(n, pathways))
    active = [kk % P for kk in range(k)]
    eng.set_active(active)
    gamma, delta = norm_x / k, 10.0 / norm_x                     # CLI defaults (prmf_runner.py:644-645)
    U = syn.init_factor_torch(torch, dev, args.seed + 1, m, k, lo, hi)
    V = syn.init_factor_torch(torch, dev, args.seed + 2, n, k)
    U0, V0 = U.clone(), V.clone()

    if world > 1:
        from prmf_amd.distributed import DistributedPrmf
        # rehearsal with the RCCL test double (PRMF_RCCL_LIB): force the in-library step although the backend is gloo
        runner = DistributedPrmf(eng, native=True if os.environ.get("PRMF_RCCL_LIB") else None)
        step = lambda s: runner.step(U, V, s, gamma, delta)
    else:
        step = lambda s: eng.step(U, V, s, gamma, delta)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
    el = torch.tensor(rep_s, dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    rep_s = sorted(float(x) for x in el.tolist())
    elapsed = rep_s[len(rep_s) // 2] if len(rep_s) % 2 else 0.5 * (rep_s[len(rep_s) // 2 - 1] + rep_s[len(rep_s) // 2])

    # live roofline of the dominant kernel (the X / X^T sweep): HIP events on the
    # launch stream around every sweep launch of a short extra run
    eng.set_timing(True)
    step(min(args.steps, 20))
    torch.cuda.synchronize()
    ms_xv, ms_xtu, _ = eng.timing()
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

    out = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.workload)
        if world == 1 and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_sweep_launch")
        out = {
            "metric": "nmf_update_iterations_per_sec", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if x_dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic X %dx%d (%s), k=%d, %d random tree pathways of 300 genes, "
                                   "active pathway k mod P, gamma=||X||/k, delta=10/||X||"
                                   % (args.workload, m, n, x_dtype, k, P),
                       "samples": m, "genes": n, "k_latent": k, "pathways": P,
                       "launches_per_iteration": (None if world > 1 else (7 if k <= 64 else 11) if args.eager_objective
                                                  else (4 if k <= 64 else 6)),
                       "parallelism": ("single GPU" if world == 1 else
                                       "X row-sharded x%d, V update %s, RCCL %s" % (
                                           world, eng.dist_mode, "in-library" if runner.native else "host-driven"))},
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
        out["reps"] = {"n": len(rep_s), "ms_per_step_median": 1e3 * elapsed / args.steps,
                       "ms_per_step_min": 1e3 * rep_s[0] / args.steps, "ms_per_step_max": 1e3 * rep_s[-1] / args.steps}
        if world > 1:
            ci = eng.comm_info()
            out["distributed"] = {
                "mode": ci["mode"], "collectives": "in-library RCCL (prmf_step_dist)" if runner.native else "host-driven (torch.distributed)",
                "rccl_comm_world_size": ci["comm_nranks"], "torch_world_size": world,
                "exchanges_per_iteration": (["one RCCL group: all_reduce U^T U (k*k f64) + reduce_scatter X^T U (n*k/N f32 per rank)",
                                             "all_gather V (n*k/N f32 per rank + the own-column update's sums)"] if ci["mode"] == "colshard" else
                                            ["one RCCL group: all_reduce U^T U (k*k f64) + all_reduce X^T U (n*k f32)"]),
                "phase_ms_rank0": dict(zip(["phase_u", "phase_q+exchange", "phase_v+exchange", "phase_post"], ci["ms_phase"]))}
        if backend != "nccl":
            out["config"]["rehearsal"] = "ranks share one GPU, exchanges staged through the host (%s): not a measurement" % backend
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch, Xd, U0, V0, pathways, active, gamma, delta, m, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
