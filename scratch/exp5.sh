export PRMF_SWEEP_NW=8 PRMF_SWEEP_BDB=1
for dbg in 0 2; do for sp in 3 6; do
  echo -n "dbg $dbg splits $sp: "; PRMF_SWEEP_DEBUG=$dbg PRMF_SWEEP_SPLITS=$sp python scratch/time_sweep.py 2>&1 | tail -1
done; done
