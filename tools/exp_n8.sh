for w in 1 2 3 4; do echo -n "wgs/cu=$w: "; PRMF_SWEEP_WGS_PER_CU=$w python tools/time_rank_share.py 8 2>&1 | tail -1; done
