# final round-2 batch on the final tree: tests -> benches -> kernel traces -> PMC traffic (separate FETCH / WRITE passes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02v; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?; echo pytest_rc=$rc
if [ $rc -ge 124 ]; then tail -3 $OUT/pytest.log | cut -c1-300; echo "pytest crashed or timed out: no further GPU step"; exit 1; fi
tail -2 $OUT/pytest.log
python -c "import torch; print(torch.zeros(4, device='cuda').sum().item())" || exit 1
for wl in c3 c2 c5shard c3f32 c5shardf32; do
  python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { echo "bench $wl failed"; exit 1; }
done
python -u tools/time_rank_share.py 8 > $OUT/rank_share8.txt 2>&1 || exit 1
for wl in c3 c2 c5shard c5shardf32; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload $wl > $OUT/prof_$wl.log 2>&1 || { echo "trace $wl failed"; exit 1; }
  python3 tools/rocpd_stats.py $(find $OUT/prof_$wl -name "*_results.db" | head -1) $OUT/kernel_stats_$wl.csv > $OUT/kernel_stats_$wl.txt 2>&1; rm -rf $OUT/prof_$wl
done
rocprofv3 --kernel-trace --stats -d $OUT/prof_rs -o rs -- python3 tools/time_rank_share.py 8 > $OUT/prof_rs.log 2>&1 && python3 tools/rocpd_stats.py $(find $OUT/prof_rs -name "*_results.db" | head -1) $OUT/kernel_stats_rankshare8.csv > /dev/null 2>&1; rm -rf $OUT/prof_rs
cd /tmp
for wl in c3 c2 c5shard c3f32 c5shardf32; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --reps 1 --workload $wl > $OUT/pmc_${wl}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --reps 1 --workload $wl > $OUT/pmc_${wl}_write.log 2>&1 || exit 1
  python3 $R/tools/pmc_summary.py $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write $OUT/pmc_traffic_${wl}.json $wl > $OUT/pmc_traffic_${wl}.txt 2>&1
  echo "$wl pmc done"; rm -rf $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write
done
grep -h -i "sweep" $OUT/pmc_traffic_*.txt | cut -c1-200
