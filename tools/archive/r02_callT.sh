OUT=gpurun_out/r02w; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -q -rA -p no:cacheprovider -x > $OUT/pytest.log 2>&1; rc=$?; echo pytest_rc=$rc; tail -3 $OUT/pytest.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" $OUT/pytest.log | head -20; echo "tests failed: no further GPU step"; exit 1; fi
python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload c5shard > $OUT/bench_c5shard.json 2> $OUT/bench_c5shard.err && \
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload c5shard > $OUT/prof_c5.log 2>&1
echo rc=$?; cut -c1-300 $OUT/bench_c5shard.json
