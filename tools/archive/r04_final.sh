# round-4 numbers on the final tree: PART=1 benches + kernel traces, PART=2 PMC traffic + MFMA counters; WLS="c5 c5shard" restricts the workloads
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
if [ "${PART:-1}" = 1 ]; then
  for wl in ${WLS:-c3 c2 c5 c3f32 c5shard c5shardf32}; do
    python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { echo "bench $wl failed"; tail -3 $OUT/bench_$wl.err; exit 1; }
    python3 -c "import json; d=json.loads(open('$OUT/bench_$wl.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$wl %.1f it/s %.4f ms | sweep %.4f/%.4f ms frac %.3f of-stream %.3f (stream %.0f GB/s) share %.3f iter_frac %.3f | update-only %.1f it/s' % (d['value'], d['ms_per_step'], r['ms_xv'], r['ms_xtu'], r['frac'], r.get('frac_of_stream', 0), r.get('stream_gbs', 0), r['sweep_share_of_step'], r['iteration_frac'], d.get('update_only', {}).get('value', 0)))"
  done
  for wl in ${WLS:-c3 c2 c5 c5shard c3f32}; do
    rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --reps 2 --workload $wl > $OUT/prof_$wl.log 2>&1 || { echo "trace $wl failed"; exit 1; }
    python3 tools/rocpd_stats.py $(find $OUT/prof_$wl -name "*_results.db" | head -1) $OUT/kernel_stats_$wl.csv > $OUT/kernel_stats_$wl.txt 2>&1; rm -rf $OUT/prof_$wl
    grep -i "sweep\|epilogue" $OUT/kernel_stats_$wl.csv | cut -c1-110
  done
else
  cd /tmp
  for wl in ${WLS:-c3 c2 c5 c3f32 c5shard}; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --reps 1 --workload $wl > $OUT/pmc_${wl}_fetch.log 2>&1 || exit 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --reps 1 --workload $wl > $OUT/pmc_${wl}_write.log 2>&1 || exit 1
    python3 $R/tools/pmc_summary.py $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write $OUT/pmc_traffic_${wl}.json $wl > $OUT/pmc_traffic_${wl}.txt 2>&1
    echo "$wl pmc done"; rm -rf $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write
  done
  grep -h -i "sweep" $OUT/pmc_traffic_*.txt | cut -c1-200
  cd $R && bash tools/pmc_mfma.sh ${WLS:-c3 c3f32 c5shard c5 c2}
fi
