mkdir -p gpurun_out/r02 && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -u -m pytest tests/test_gpu_solver.py -m gpu -q -rA -p no:cacheprovider -k "outer_solver or cli_outputs" > gpurun_out/r02/pytest2.log 2>&1; echo pytest_rc=$?
./build_variants/mall_reread > gpurun_out/r02/mall_reread.txt 2>&1; echo mall_rc=$?
python -u tools/time_call_overhead.py > gpurun_out/r02/call_overhead_c3.txt 2>&1; echo ovh_rc=$?
WORKLOAD=c2 python -u tools/time_call_overhead.py > gpurun_out/r02/call_overhead_c2.txt 2>&1
python -u bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_c3_a.json 2> gpurun_out/r02/bench_c3_a.err; echo bench_rc=$?
python -u bench.py --steps 20 --warmup 5 --workload c2 > gpurun_out/r02/bench_c2_a.json 2> gpurun_out/r02/bench_c2_a.err
python -u tools/time_rank_share.py 8 > gpurun_out/r02/rank_share8.txt 2>&1
MODE=colshard python -u tools/time_rank_share.py 8 >> gpurun_out/r02/rank_share8.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/prof_c3 -o c3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 > gpurun_out/r02/prof_c3.log 2>&1; echo prof_rc=$?
rocprofv3 --kernel-trace --stats -d gpurun_out/r02/prof_c2 -o c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload c2 > gpurun_out/r02/prof_c2.log 2>&1
ls gpurun_out/r02 gpurun_out/r02/prof_c3 | head -40
