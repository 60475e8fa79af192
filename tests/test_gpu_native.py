"""The C-ABI driven from plain C++ (tests/native/capi_native.cpp: hipMalloc buffers, no Python or torch in the
process) against the oracle on the same LCG-generated inputs."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lcg_stream(count, state=12345):
    out = np.empty(count)
    for i in range(count):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = ((state >> 8) + 1) / 16777216.0
    return out


def test_native_driver_matches_oracle(tmp_path):
    from oracle import prmf_oracle as orc
    from prmf_amd import _build
    lib = _build.build()
    exe = str(tmp_path / "capi_native")
    libdir = os.path.dirname(lib)
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "native", "capi_native.cpp"), "-o", exe, "-L", libdir, "-lprmf_hip",
                         "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "native OK" in r.stdout
    m, n, k, P, S = 70, 150, 4, 3, 12
    s = lcg_stream(m * n + m * k + n * k)
    X = s[:m * n].astype(np.float32).astype(np.float64).reshape(m, n)
    U0 = (3.0 * s[m * n:m * n + m * k]).astype(np.float32).astype(np.float64).reshape(m, k)
    V0 = (3.0 * s[m * n + m * k:]).astype(np.float32).astype(np.float64).reshape(n, k)
    pws = []
    for p in range(P):
        nodes = np.arange(p * 20, p * 20 + S)
        edges = np.stack([nodes[:-1], nodes[1:]], axis=1)
        pws.append(orc.pathway_from_edges(n, edges, nodes, np.ones(len(edges))))
    tr = []
    Ur, Vr, od = orc.nmf_manifold_vec_update(X, U0, V0, [pws[0], pws[1], pws[2], pws[0]], n_steps=3, gamma=2.5, delta=0.125,
                                             trace=tr)
    lines = [l.split() for l in r.stdout.splitlines()]
    got = [[float(x) for x in l[2:]] for l in lines if l and l[0] == "obj"]
    assert len(got) == 3
    for g, t in zip(got, tr):
        want = [t["recon"], t["manifold"], t["ignore"], t["fro"], t["obj"]]
        assert np.allclose(g, want, rtol=1e-4)
    sums = [l for l in lines if l and l[0] == "sumU"][0]
    assert abs(float(sums[1]) - Ur.sum()) < 1e-4 * Ur.sum() and abs(float(sums[3]) - Vr.sum()) < 1e-4 * Vr.sum()


def _build_native(src, exe, libdir, extra=()):
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "native", src), "-o", exe, "-L", libdir, "-lprmf_hip",
                         "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"] + list(extra),
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    return exe


def test_native_driver_covers_the_rest_of_the_api(tmp_path):
    """tests/native/capi_wide.cpp, torch-free: nine contexts of different plans (f32 / bf16 X, k = 1 .. 128, 4- and 8-wave
    sweeps, ragged sizes), prmf_step with the tradeoff feedback and in the eager / direct-residual forms, the phase calls,
    candidate scores, held-out error, direct residual, timing, introspection, the quantile transform, invalidation and the
    error paths on a live context -- every product entry point of include/prmf_hip.h from plain C++."""
    from prmf_amd import _build
    libdir = os.path.dirname(_build.build())
    exe = _build_native("capi_wide.cpp", str(tmp_path / "capi_wide"), libdir)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "wide OK" in r.stdout and r.stdout.count(": ok (") == 9


def test_native_multi_rank_driver(tmp_path):
    """tests/native/capi_ranks.cpp, torch-free: the in-library multi-rank step on 2 .. 8 rank THREADS of one C++ process
    sharing the GPU (collectives through the RCCL double), both V-update modes, ranks without rows / without a latent
    column: identical V and records on every rank, U / V / objective within 1e-5 of the single-rank step on the whole matrix."""
    from prmf_amd import _build
    libdir = os.path.dirname(_build.build())
    mock = str(tmp_path / "libmock_rccl.so")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                         os.path.join(ROOT, "tests", "native", "mock_rccl.cpp"), "-o", mock, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    exe = _build_native("capi_ranks.cpp", str(tmp_path / "capi_ranks"), libdir)
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, PRMF_RCCL_LIB=mock), timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ranks OK" in r.stdout and r.stdout.count("ranks ") >= 6
