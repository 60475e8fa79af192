# refresh of the k = 128 numbers after the 8-wave X^T U shape: benches, kernel trace + PMC traffic of c5, MFMA counters
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03f2; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for wl in c5 c5shard c3 c2; do
  python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { echo "bench $wl failed"; exit 1; }
  python3 -c "import json; d=json.loads(open('$OUT/bench_$wl.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$wl %.1f it/s %.4f ms | sweep %.4f/%.4f ms frac %.3f share %.3f iter_frac %.3f' % (d['value'], d['ms_per_step'], r['ms_xv'], r['ms_xtu'], r['frac'], r['sweep_share_of_step'], r['iteration_frac']))"
done
for wl in c5 c5shard; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload $wl > $OUT/prof_$wl.log 2>&1 || { echo "trace $wl failed"; exit 1; }
  python3 tools/rocpd_stats.py $(find $OUT/prof_$wl -name "*_results.db" | head -1) $OUT/kernel_stats_$wl.csv > $OUT/kernel_stats_$wl.txt 2>&1; rm -rf $OUT/prof_$wl
  grep -i "sweep\|epilogue\|prologue" $OUT/kernel_stats_$wl.csv | cut -c1-110
done
cd /tmp
for wl in c5; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --reps 1 --workload $wl > $OUT/pmc_${wl}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --reps 1 --workload $wl > $OUT/pmc_${wl}_write.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write $OUT/pmc_traffic_${wl}.json $wl > $OUT/pmc_traffic_${wl}.txt 2>&1
  rm -rf $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write
done
grep -h -i "sweep" $OUT/pmc_traffic_*.txt | cut -c1-200
cd $R && bash tools/pmc_mfma.sh c5 c5shard
