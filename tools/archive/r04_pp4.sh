R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
L=prmf_amd/lib/variants/libprmf_hip
for v in pp1k64w; do
PRMF_HIP_LIB=$R/${L}_$v.so timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_medium.py tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -m gpu > $OUT/pytest_$v.log 2>&1; rc=$?; tail -3 $OUT/pytest_$v.log
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest_$v.log | head; exit $rc; }
done
ROUNDS=5 timeout -k 10 1000 python tools/ab_sweep.py c3,c4share8,c4share2 base=prmf_amd/lib/libprmf_hip.so pp1k64=${L}_pp1k64.so pp1k64w=${L}_pp1k64w.so > $OUT/ab_pp4.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_pp4.txt || tail -5 $OUT/ab_pp4.txt
exit $rc
