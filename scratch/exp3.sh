for nw in 4 8; do for bdb in 0 1; do
  export PRMF_SWEEP_NW=$nw PRMF_SWEEP_BDB=$bdb
  echo "== NW=$nw BDB=$bdb"; python scratch/race_probe.py 2>&1 | grep -E "differs|FPE" | head -4
  for sp in 0 2 3 4 6 8; do
    if [ $sp -gt 0 ]; then export PRMF_SWEEP_SPLITS=$sp; else unset PRMF_SWEEP_SPLITS; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  splits', '$sp', round(d['value'],1), round(d['roofline']['ms_xv'],4), round(d['roofline']['ms_xtu'],4))
except Exception as e: print('  splits $sp FAILED')"
  done; unset PRMF_SWEEP_SPLITS
done; done
