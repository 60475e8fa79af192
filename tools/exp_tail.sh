python -m pytest tests/test_gpu_parity.py tests/test_gpu_medium.py -m gpu -x -q 2>&1 | tail -2
python tools/race_probe.py 2>&1 | grep -E "differs|FPE" | head -3
for rep in 1 2; do for t in 0 6 12 20; do echo -n "tail=$t%: "; PRMF_SWEEP_TAIL=$t python tools/time_sweep.py 2>&1 | tail -1; done; done
