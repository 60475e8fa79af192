# round-2 GPU batch: tests, then (only if the GPU is still healthy) benches and kernel traces
OUT=gpurun_out/r02y; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -q -rA -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?; echo pytest_rc=$rc; if [ $rc -ge 124 ]; then tail -3 $OUT/pytest.log | cut -c1-300; echo "pytest crashed or timed out: no further GPU step"; exit 1; fi
tail -4 $OUT/pytest.log
python -c "import torch; print(torch.zeros(4, device='cuda').sum().item())" || exit 1
grep -q "failed" $OUT/pytest.log && echo "TESTS FAILED (benches still run)"
python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err && \
python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err && \
python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload c5shard > $OUT/bench_c5shard.json 2> $OUT/bench_c5shard.err && \
python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload c3f32 > $OUT/bench_c3f32.json 2> $OUT/bench_c3f32.err && \
python -u tools/time_rank_share.py 8 > $OUT/rank_share8.txt 2>&1 && \
rocprofv3 --kernel-trace --stats -d $OUT/prof_c3 -o c3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 > $OUT/prof_c3.log 2>&1 && \
rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload c2 > $OUT/prof_c2.log 2>&1 && \
rocprofv3 --kernel-trace --stats -d $OUT/prof_rs -o rs -- python3 tools/time_rank_share.py 8 > $OUT/prof_rs.log 2>&1
echo chain_rc=$?
cat $OUT/bench_c3.json $OUT/bench_c2.json | cut -c1-400
