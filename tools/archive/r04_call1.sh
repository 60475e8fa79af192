# round 4, first GPU call: the new multi-rank tests (8 thread ranks through the RCCL double), the old multi-rank tests on the
# rank-group protocol, and the new bench line (stream calibration, update-only, outer loop, full-size CPU baseline)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c1; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout -k 10 1500 python -m pytest tests/test_gpu_eightway.py -x -q -m gpu -rP > $OUT/eightway.log 2>&1; rc=$?; tail -5 $OUT/eightway.log; grep "measured" $OUT/eightway.log | cut -c1-260
[ $rc = 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_sharded_solver.py tests/test_gpu_recon.py -x -q -m gpu > $OUT/dist.log 2>&1; rc=$?; tail -3 $OUT/dist.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python -u bench.py --workload c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; rc=$?; tail -2 $OUT/bench_c3.err; cut -c1-1500 $OUT/bench_c3.json
exit $rc
