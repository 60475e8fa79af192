# kernel trace of bench.py for the given workloads -> gpurun_out/r03t/kernel_stats_<wl>.csv (+ .txt with the gaps)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03t; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for wl in "$@"; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload $wl > $OUT/prof_$wl.log 2>&1 || { echo "trace $wl failed"; tail -3 $OUT/prof_$wl.log; exit 1; }
  python3 tools/rocpd_stats.py $(find $OUT/prof_$wl -name "*_results.db" | head -1) $OUT/kernel_stats_$wl.csv > $OUT/kernel_stats_$wl.txt 2>&1; rm -rf $OUT/prof_$wl
  echo "== $wl"; head -8 $OUT/kernel_stats_$wl.csv | cut -c1-120; tail -12 $OUT/kernel_stats_$wl.txt | cut -c1-160
done
