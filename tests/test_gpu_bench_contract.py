"""bench.py's ONE JSON line: the keys and relations the driver and the judge read (task contract: metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, plus the
`roofline` and `cpu_baseline` objects) -- on the small BASELINE configs[1] workload with a short CPU-baseline budget, so
that the check itself takes seconds.  The default invocation (C3, the headline) differs only in the workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c2", "--steps", "6", "--warmup", "2", "--reps", "3",
                        "--cpu-seconds", "3", "--secondary", "c2,c4_share_of_8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout: %r" % lines[:3]
    d = json.loads(lines[0])
    assert d["metric"] == "nmf_update_iterations_per_sec" and d["unit"] == "iterations/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] in ("weak", "strong")
    assert "5000x10000" in d["config"]["workload"] and d["config"]["k_latent"] == 16 and d["config"]["launches_per_iteration"] == 4
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 1e-6                      # value = iterations / second of the timed block
    assert d["recon_err"] > 0 and d["objective"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    # both bounds are priced; `bound` names the larger floor; the top-level numbers are the binding bound's
    hb, mf = rf["hbm"], rf["mfma"]
    assert hb["floor_ms"] > mf["floor_ms"] and rf["achieved"] == hb["achieved"] and rf["frac"] == hb["frac"]
    assert mf["mfma_per_product"] == 5 and mf["k_padded"] == 32                                # f32 X: three pieces x hi/lo, k = 16 padded to 32
    assert mf["issued_flops_per_launch"] == 5 * 2.0 * 5000 * 10000 * 32
    assert abs(mf["achieved"] - mf["issued_flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * mf["achieved"]
    assert abs(mf["frac_of_nominal"] - mf["achieved"] / 2500.0) < 1e-12 and mf["peak_nominal"] == 2500.0
    assert (mf["clock_ghz"] is None and mf["peak"] == 2500.0) or abs(mf["peak"] - 2500.0 * mf["clock_ghz"] / 2.4) < 1e-9
    assert abs(hb["floor_ms"] - 1e3 * rf["algorithmic_bytes_per_launch"] / 8e12) < 1e-12
    assert rf["traffic"] is None or abs(rf["traffic_ratio"] - rf["traffic"] / rf["algorithmic_bytes_per_launch"]) < 1e-12
    assert len(rf["traffic_source"]["sweep_sources_sha16"]) == 16
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.2 < rf["frac"] < 1.0
    # achieved = algorithmic bytes per launch / average launch duration measured in this run (HIP events on the launch stream)
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["algorithmic_bytes_per_launch"] == 5000 * 10000 * 4 + 4 * 10000 * 16 + 4 * 5000 * 16      # X once + factor once + product once
    assert rf["traffic"] is None or rf["traffic"] > 0.9 * rf["algorithmic_bytes_per_launch"]
    assert 0 < rf["frac_of_stream"] <= 1.05 and rf["stream_gbs"] > 1000
    assert 2 * rf["avg_launch_ms"] < d["ms_per_step"] * 1.02                                 # the two sweeps fit inside the step
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "iterations/s" and cb["value"] > 0 and cb["cores"] >= 1 and "oracle/prmf_oracle.py" in cb["sample"]
    assert cb["method"]["rebuild_laplacians"] is True and cb["lean_value"] >= cb["value"] * 0.5
    assert d["update_only"]["value"] > 0 and d["outer_loop"]["iterations"] == 100
    ol = d["outer_loop"]
    assert abs(ol["it_per_s_warm"] * ol["seconds_warm_call"] - 100) < 1e-6 and abs(ol["it_per_s_device_resident"] * ol["seconds_device_resident"] - 100) < 1e-6
    # the reported warm / device-resident call is the median of the three that ran, all of them listed
    assert sorted(ol["seconds_warm_calls"])[1] == ol["seconds_warm_call"] and sorted(ol["seconds_device_resident_calls"])[1] == ol["seconds_device_resident"]
    # the other workloads of the same run: the same arithmetic
    sec = d["secondary"]
    assert sorted(sec) == ["c2", "c4_share_of_8"]
    s2 = sec["c2"]
    assert abs(s2["it_per_s"] * s2["ms_per_step"] / 1000.0 - 1.0) < 1e-6 and s2["bound"] == "hbm"
    assert s2["algorithmic_bytes_per_launch"] == rf["algorithmic_bytes_per_launch"]
    assert abs(s2["hbm_frac"] - s2["algorithmic_bytes_per_launch"] / (0.5e-3 * sum(s2["sweep_ms"])) / 1e9 / 8000.0) < 1e-9
    assert sum(s2["sweep_ms"]) < s2["ms_per_step"] * 1.02 and 0.2 < s2["hbm_frac"] < 1.0
    sh = sec["c4_share_of_8"]
    assert "6250 rows" in sh["workload"] and sh["speedup_bound_before_exchanges"] is None          # (no c3 headline in this run to divide by)
    assert sum(sh["sweep_ms"]) < sh["ms_per_iteration_device"] and abs(sh["other_kernels_ms"] - (sh["ms_per_iteration_device"] - sum(sh["sweep_ms"]))) < 1e-9
    # the stand-in copies, timed alone, are not part of the kernels' figure
    assert 0 < sh["stand_in_copies_ms"] < 0.05 and abs(sh["ms_per_iteration_kernels"] - (sh["ms_per_iteration_device"] - sh["stand_in_copies_ms"])) < 1e-12
    assert abs(sh["iteration_frac"] - (2.0 * 6250 * 20000 * 2 + 12.0 * (6250 + 20000) * 64) / (sh["ms_per_iteration_kernels"] * 1e-3) / 1e9 / 8000.0) < 1e-9
    assert abs(sh["sweep_hbm_frac"] - (6250 * 20000 * 2 + 2.0 * (20000 + 6250) * 64) / (0.5e-3 * sum(sh["sweep_ms"])) / 1e9 / 8000.0) < 1e-9
