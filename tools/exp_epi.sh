python -m pytest tests -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof5.log 2>&1
grep -o '"value": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/prof5.log
