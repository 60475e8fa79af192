for pad in 0 64 128 192 256 320 448 576 1088 2112; do echo -n "pad $pad: "; PRMF_LD_PAD=$pad PRMF_SWEEP_DEBUG=2 python scratch/time_sweep.py 2>&1 | tail -1; done
