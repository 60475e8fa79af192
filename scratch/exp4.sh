export PRMF_SWEEP_NW=8 PRMF_SWEEP_BDB=1
for dbg in 0 1 2; do for sp in 3 6; do
  export PRMF_SWEEP_DEBUG=$dbg PRMF_SWEEP_SPLITS=$sp
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('dbg $dbg splits $sp', round(d['value'],1), round(d['roofline']['ms_xv'],4), round(d['roofline']['ms_xtu'],4))
except Exception as e: print('dbg $dbg splits $sp FAILED')"
done; done
