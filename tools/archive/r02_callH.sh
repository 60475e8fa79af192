OUT=gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -q -rA -p no:cacheprovider -x > $OUT/pytest.log 2>&1; rc=$?; echo pytest_rc=$rc; tail -3 $OUT/pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then echo "tests failed: no further GPU step"; exit 1; fi
WORKLOAD=c3 python -u tools/sweep_knobs.py > $OUT/knobs_c3.txt 2>&1 && \
WORKLOAD=c2 python -u tools/sweep_knobs.py > $OUT/knobs_c2.txt 2>&1 && \
WORKLOAD=c3f32 python -u tools/sweep_knobs.py > $OUT/knobs_c3f32.txt 2>&1 && \
python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo rc=$?; cat $OUT/knobs_*.txt | grep -v amdgpu.ids; cut -c1-250 $OUT/bench_c3.json
