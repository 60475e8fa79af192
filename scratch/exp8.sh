for dbg in 0 1 2; do echo -n "dbg $dbg: "; PRMF_SWEEP_DEBUG=$dbg python scratch/time_sweep.py 2>&1 | tail -1; done
PRMF_SWEEP_DEBUG=2 python scratch/race_probe.py 2>&1 | grep -E "differs|FPE|relerr" | head -8
