"""Build libprmf_hip.so (gfx950) in-tree with hipcc.  No torch involved."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libprmf_hip.so")
SOURCES = ["gemm.hip", "setx.hip", "update.hip", "residual.hip", "nnls.hip", "nnls_matrix.hip", "prep.hip", "textio.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I/opt/rocm/include"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libprmf_hip.so cannot be built")
    return exe


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "prmf_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


SWEEP_SOURCES = ["gemm.hip", "prologue.h", "kernels.h", "bf16split.h"]


def sweep_sources_sha16():
    """Fingerprint of the sources the sweep kernel is built from.  Counter measurements kept under profiles/ (HBM traffic,
    the clock held under the kernel) carry it, and bench.py reports them only while it still matches: the kernel cannot
    change without this changing."""
    import hashlib
    h = hashlib.sha256()
    for f in SWEEP_SOURCES:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile every translation unit for gfx950 and link the shared library."""
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr)
    return LIB


def build_variant(name, defines):
    """Experimental build with extra -D flags -> prmf_amd/lib/variants/libprmf_hip_<name>.so (tools/ab_sweep.py)."""
    vdir = os.path.join(LIBDIR, "variants")
    objdir = os.path.join(vdir, "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    out = os.path.join(vdir, "libprmf_hip_%s.so" % name)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        r = subprocess.run([cc] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr)
    return out


def build_sanitized(outdir):
    """libprmf_hip_san.so in `outdir`: the same sources with the HOST code under AddressSanitizer + UndefinedBehaviorSanitizer
    (-fsanitize=address,undefined -fno-gpu-sanitize: device code is compiled as always -- GPU sanitizers are not available
    on this pool).  Test infrastructure (tests/test_sanitize_cpu.py, tests/test_gpu_native.py); link the driver with the same
    compiler and flags so that both share one sanitizer runtime."""
    os.makedirs(outdir, exist_ok=True)
    cc = hipcc()
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined"]
    # third-party headers are not ours to fix: rocPRIM's temporary-storage size query (segmented_radix_sort_keys with a null
    # buffer, prep.hip) partitions offsets from a null base pointer -- "applying non-zero offset to null pointer"
    ignore = os.path.join(outdir, "sanitize_ignorelist.txt")
    with open(ignore, "w") as fh:
        fh.write("src:*/include/rocprim/*\nsrc:*/include/hipcub/*\n")
    flags = [f for f in FLAGS if f != "-O3"] + ["-O1", "-g"] + san + ["-fsanitize-ignorelist=" + ignore]

    def compile_one(src):
        obj = os.path.join(outdir, src.replace(".hip", ".san.o"))
        r = subprocess.run([cc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (sanitized) failed for %s:\n%s" % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    out = os.path.join(outdir, "libprmf_hip_san.so")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + san[:2] + ["-o", out] + objs + ["-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link (sanitized) failed:\n%s" % r.stderr)
    return out


def build_sanitized_driver(src, exe, libdir, extra=()):
    """a C++ driver against libprmf_hip_san.so, compiled by the same clang with the same sanitizers (one runtime for both)"""
    cc = hipcc()
    r = subprocess.run([cc, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(HERE, "..", "include"), "-I/opt/rocm/include", "-x", "c++", src, "-o", exe,
                        "-L", libdir, "-lprmf_hip_san", "-Wl,-rpath," + libdir] + list(extra), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("sanitized driver build failed:\n%s" % r.stderr)
    return exe


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # python -m prmf_amd._build --variant NAME DEF1=1 DEF2=0 ...
        print(build_variant(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
