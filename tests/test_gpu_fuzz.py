"""Seeded random shapes (a bounded run of tools/fuzz_shapes.py and tools/fuzz_ranks.py, each in a process of its own):
tiny and ragged matrices, k = 1 .. 128, both storages of X, the tradeoff feedback, candidate scores -- against the float64
oracle at 1e-4 -- and the same kind of shapes through the in-library multi-rank step on 2 .. 8 thread ranks (RCCL test
double; ranks without rows, ranks without a latent column) against the single-rank step at 1e-5, bit-identical across
ranks and runs; since round 6 also whole solves (tools/fuzz_outer.py) and the upload paths (tools/fuzz_setx.py).  The long runs
(hundreds of cases per seed) are in profiles/r05_fuzz_*.txt and profiles/r06_fuzz_*.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, seed, cases, timeout, **extra):
    env = dict(os.environ, SEED=str(seed), CASES=str(cases), **extra)
    env.pop("PRMF_RCCL_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], env=env, capture_output=True, text=True, timeout=timeout)
    tail = "\n".join((r.stdout + "\n" + r.stderr).strip().splitlines()[-12:])
    assert r.returncode == 0, tail
    assert ("all %d " % cases) in r.stdout, tail


@pytest.mark.slow
def test_random_shapes_vs_oracle():
    _run("fuzz_shapes.py", seed=11, cases=40, timeout=600)


@pytest.mark.slow
def test_random_shapes_through_the_multi_rank_step():
    # ANY_K: every k <= 128 against every rank count (the k = 96 / five-rank defect of round 6 sat between the usual suspects)
    _run("fuzz_ranks.py", seed=301, cases=30, timeout=600, ANY_K="1")


@pytest.mark.slow
def test_random_whole_solves_vs_the_oracle():
    """nmf_pathway on random problems against the oracle's nmf_pathway on the same RNG stream: the same decisions in every round
    (sampling, `restrict`, matching, stopping), trace / U / V at 1e-4; where the reference's tradeoff feedback runs away the product
    stops with its numeric error."""
    _run("fuzz_outer.py", seed=1, cases=12, timeout=600)


def test_random_shapes_through_every_upload_path():
    _run("fuzz_setx.py", seed=1, cases=8, timeout=600)


def test_random_and_mutated_text_files_vs_pandas():
    """A bounded run of tools/fuzz_text.py: well-formed and mutated delimited files through the GPU text reader -- whatever it
    returns must be pandas' frame exactly; refusing is always allowed (long runs: profiles/r05_fuzz_text.txt)."""
    env = dict(os.environ, SEED="7", CASES="600", PRMF_TEXT_READER="gpu")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_text.py")], env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + "\n" + r.stderr).strip().splitlines()[-6:])
    assert r.returncode == 0 and "accepted and equal to pandas' frame" in r.stdout, tail
    accepted = int(r.stdout.split(" files: ")[1].split(" accepted")[0])
    assert accepted > 150, tail          # the run is not vacuous


def test_random_nnls_problems_judged_by_their_optimality_conditions():
    """A bounded run of tools/fuzz_nnls.py: prmf_nnls on random problems of both orientations (dense, signed, low-rank, duplicated,
    badly scaled; right-hand sides inside the cone, with zeros, negative) -- every answer satisfies the optimality conditions and is
    at least as good as scipy's; the few it cannot resolve are declined, never answered wrongly (long runs: profiles/r05_fuzz_text.txt)."""
    env = dict(os.environ, SEED="3", CASES="150")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_nnls.py")], env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + "\n" + r.stderr).strip().splitlines()[-6:])
    assert r.returncode == 0 and "all 150 problems: optimality conditions hold" in r.stdout, tail
