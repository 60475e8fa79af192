R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
L=prmf_amd/lib/variants/libprmf_hip
PRMF_HIP_LIB=$R/${L}_ppf.so timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_medium.py tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -m gpu > $OUT/pytest_ppf.log 2>&1; rc=$?; tail -3 $OUT/pytest_ppf.log
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest_ppf.log | head; exit $rc; }
ROUNDS=5 timeout -k 10 1100 python tools/ab_sweep.py c3,c4share4,c4share8,c3f32 base=prmf_amd/lib/libprmf_hip.so ppa=${L}_ppa.so ppf=${L}_ppf.so ppr8=${L}_ppr8.so > $OUT/ab_pp5.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_pp5.txt || tail -5 $OUT/ab_pp5.txt
exit $rc
