export PRMF_SWEEP_NW=8 PRMF_SWEEP_BDB=1
for w in 2 5 8 10 12 16; do export PRMF_SWEEP_WGS_PER_CU=$w
  echo -n "NW=8 BDB=1 WGS/CU=$w: "; python scratch/time_sweep.py 2>&1 | tail -1
done
