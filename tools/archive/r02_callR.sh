OUT=gpurun_out/r02r; mkdir -p $OUT; export TMPDIR=/tmp
ab=$OUT/ab_eager_vs_deferred.txt; : > $ab
for wl in c3 c2; do for rep in 1 2; do for mode in "" "--eager-objective"; do
  python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl $mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-4s %-18s %8.1f it/s  %.4f ms/step (min %.4f max %.4f)  sweeps %.4f + %.4f ms  share %.3f  launches/iteration %s' % (
      '$wl', '$mode' or 'deferred(default)', d['value'], d['ms_per_step'], d['reps']['ms_per_step_min'], d['reps']['ms_per_step_max'], r['ms_xv'], r['ms_xtu'], r['sweep_share_of_step'], d['config']['launches_per_iteration']))" >> $ab || exit 1
done; done; done
cat $ab
