"""The code the 8-GPU launch executes, run once on the one GPU of the test box with the REAL librccl.

`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the driver's command) and `prmf_runner.py` under the
same launcher take a branch no single-GPU test reaches: `dist.init_process_group("nccl", device_id=...)`, the host-side rank
protocol over NCCL tensors (prmf_amd.ranks.TorchRanks), `prmf_comm_unique_id` -> broadcast -> `prmf_comm_init` on librccl,
`prmf_step_dist`.  RCCL refuses two ranks on one device, so the branch is taken with ONE process: PRMF_FORCE_RANK_PATH=1
makes a one-rank process group count as a rank group (its collectives are identities, its call sequence is the real one).
Every run below is a fresh child process started before anything in it has touched the GPU.  No reference counterpart
(script/prmf_runner.py:419-449 is one process); the oracle here is the single-process path of the same build: the results
must be IDENTICAL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from test_gpu_distributed import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script_args, cwd, force=True, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("PRMF_RCCL_LIB", None)                    # the real RCCL, not the test double
    if force:
        env["PRMF_FORCE_RANK_PATH"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, env=env, timeout=timeout)


def _bench_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_the_launcher_runs_the_rccl_process_path(tmp_path):
    args = ["--gpus", "1", "--steps", "12", "--warmup", "3", "--reps", "2", "--workload", "c2", "--no-cpu-baseline", "--no-extras"]
    forced = _bench_line(_launch([os.path.join(ROOT, "bench.py")] + args, str(tmp_path)))
    plain = _bench_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=str(tmp_path), capture_output=True,
                                       text=True, timeout=900))
    d = forced["distributed"]
    assert d["rccl_comm_world_size"] == 1 and d["world_size"] == 1
    assert d["collectives"].startswith("in-library RCCL") and d["ranks_are"].startswith("processes")
    assert "config" in forced and "rehearsal" not in forced["config"]
    assert "distributed" not in plain
    # prmf_step_dist on a one-rank RCCL communicator == prmf_step, bit for bit: same solve, same records
    assert forced["recon_err"] == plain["recon_err"] and forced["objective"] == plain["objective"]
    assert forced["n_gpus"] == 1 and forced["steps"] == 12 and forced["value"] > 0


@pytest.mark.parametrize("norm", ["nonorm", "norm"])
def test_cli_under_the_launcher_equals_the_single_process_run(tmp_path, norm):
    """script/prmf_runner.py as rank 0 of a one-process NCCL group (rank-local .npy input: runner.RowSource, the
    column-sharded normalisation, nmf_pathway(group=world) with rank 0's draws broadcast, the in-library RCCL step) against
    the same command started plainly: U.csv, V.csv, obj.txt byte for byte."""
    c = synth.cli_case(str(tmp_path), variant=1)
    npy = str(tmp_path / "data.npy")
    np.save(npy, c["X"])
    outs = {}
    for how in ("forced", "plain"):
        out = str(tmp_path / ("out_" + how))
        os.makedirs(out)
        args = [os.path.join(ROOT, "script", "prmf_runner.py"), "--data", npy, "--manifolds"] + c["manifolds"] + [
            "--node-attribute", "name", "--nodelist", c["nodelist"], "--outdir", out, "--seed", "1"] + (["--no-normalize"] if norm == "nonorm" else [])
        if how == "forced":
            r = _launch(args, str(tmp_path))
        else:
            r = subprocess.run([sys.executable] + args, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[how] = (out, r.stdout)
    for name in ("U.csv", "V.csv", "obj.txt"):
        a = open(os.path.join(outs["forced"][0], name), "rb").read()
        b = open(os.path.join(outs["plain"][0], name), "rb").read()
        assert a == b, name
    keep = lambda text: [l for l in text.splitlines() if l.split() and l.split()[0].isdigit()]
    assert keep(outs["forced"][1]) == keep(outs["plain"][1])
