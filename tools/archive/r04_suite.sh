# the whole GPU suite + smoke on the tree as it stands; the errors every test measured go to gpurun_out/parity_measured.jsonl
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04s; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout -k 10 1700 python -m pytest ${PYTEST_ARGS:-tests} -x -q -m gpu --durations=12 > $OUT/pytest.log 2>&1; rc=$?; tail -22 $OUT/pytest.log
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest.log | head -20; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -3 $OUT/smoke.log
exit $rc
