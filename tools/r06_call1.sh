# round 6, call 1: the new upload paths, the forced rank-process path, the outer loop by stage
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
python -m pytest tests/test_gpu_setx.py tests/test_gpu_rank_process.py -m gpu -x -q > $OUT/t1.txt 2>&1; echo "new tests rc=$?"; tail -15 $OUT/t1.txt
python -m pytest tests/test_gpu_solver.py tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -x -q > $OUT/t2.txt 2>&1; echo "solver tests rc=$?"; tail -5 $OUT/t2.txt
TOP=30 python tools/profile_outer_loop.py > $OUT/outer_profile.txt 2>&1; echo "profile rc=$?"; grep -v "^ \|^$" $OUT/outer_profile.txt | head -40
python bench.py --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$OUT/bench_c3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['outer_loop'], indent=1))"
