# ping-pong default: whole GPU suite, then same-session A/B against the -DPRMF_PP=0 build, then a short bitwise soak
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout -k 10 1700 python -m pytest tests -x -q -m gpu > $OUT/pytest_default.log 2>&1; rc=$?; tail -3 $OUT/pytest_default.log
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest_default.log | head -20; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUNDS=7 timeout -k 10 1100 python tools/ab_sweep.py c3,c4share2 pp=prmf_amd/lib/libprmf_hip.so nopp=prmf_amd/lib/variants/libprmf_hip_nopp.so > $OUT/ab_pp6.txt 2>&1; rc=$?
grep -v amdgpu $OUT/ab_pp6.txt | tail -30
[ $rc = 0 ] || exit $rc
WORKLOAD=c3 REPS=60 timeout -k 10 600 python tools/soak_determinism.py 2>&1 | tail -3
