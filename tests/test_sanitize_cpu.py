"""Host side of the C-ABI under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are
not available on this pool): the library is rebuilt with both sanitizers on its HOST code only (prmf_amd._build.build_sanitized; device code as
always) and tests/native/host_sanitize.cpp drives what runs without a GPU -- the sweep planner over 300 000 random
shapes (decomposition, slab masks: 64-bit shifts, plan vectors) and the argument / error paths in front of any device
work.  Any sanitizer report aborts the driver (-fno-sanitize-recover)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_planner_and_error_paths_under_asan_ubsan(tmp_path):
    from prmf_amd import _build
    libdir = str(tmp_path / "san")
    _build.build_sanitized(libdir)
    exe = _build.build_sanitized_driver(os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), str(tmp_path / "host_sanitize"), libdir)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, "300000"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "300000 plans checked, 0 failures" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
