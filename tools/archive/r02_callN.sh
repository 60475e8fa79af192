# profiles batch: PMC traffic per workload (separate FETCH / WRITE passes), kernel stats of the widened rows
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02n; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for wl in c3 c2 c5shard c3f32; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --reps 1 --workload $wl > $OUT/pmc_${wl}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --reps 1 --workload $wl > $OUT/pmc_${wl}_write.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write $OUT/pmc_traffic_${wl}.json $wl > $OUT/pmc_traffic_${wl}.txt 2>&1
  echo "$wl pmc done"; rm -rf $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write
done
for t in time_score time_cv_error time_quantile; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$t -o $t -- python3 $R/tools/$t.py > $OUT/$t.log 2>&1 || exit 1
  echo "$t done"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_score_fetch -- python3 $R/tools/time_score.py > $OUT/pmc_score_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_score_write -- python3 $R/tools/time_score.py > $OUT/pmc_score_write.log 2>&1 && \
python3 $R/tools/pmc_summary.py $OUT/pmc_score_fetch $OUT/pmc_score_write $OUT/pmc_traffic_score.json score > $OUT/pmc_traffic_score.txt 2>&1
rm -rf $OUT/pmc_score_fetch $OUT/pmc_score_write
cat $OUT/pmc_traffic_c3.txt | grep -i "sweep\|epilogue"
