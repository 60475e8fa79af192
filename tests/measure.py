"""Every GPU parity test reports the error it measured, not only pass/fail: printed (pytest -s / -rP shows it)
and appended as one JSON line to $PRMF_MEASURE_LOG (default gpurun_out/parity_measured.jsonl when that directory
exists), so the tolerances in the tests can be read against the headroom they leave."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    """||a - b||_F / ||b||_F in float64."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def relmax(a, b):
    """max over entries of |a - b| / max(|b|, tiny): for short vectors of scalars (objective components, traces)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def report(test, **values):
    rec = {"test": test}
    rec.update({k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in values.items()})
    print("[measured] " + json.dumps(rec))
    path = os.environ.get("PRMF_MEASURE_LOG")
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        path = os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl")
    if path:
        try:
            with open(path, "a") as fh:
                fh.write(json.dumps(rec) + "\n")
        except OSError:
            pass
    return rec
