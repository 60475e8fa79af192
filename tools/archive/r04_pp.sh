# ping-pong chunk loop (variant build -DPRMF_PP=1): parity at k > 64, then same-session A/B against the library as built
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
V=${VARIANT:-pp}
PRMF_HIP_LIB=$R/prmf_amd/lib/variants/libprmf_hip_$V.so timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_medium.py -x -q -m gpu -k "${KEXPR:-c5 or tallthin128 or 128 or 100 or 96}" > $OUT/pytest_$V.log 2>&1; rc=$?; tail -4 $OUT/pytest_$V.log
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest_$V.log | head; exit $rc; }
ROUNDS=${ROUNDS:-3} timeout -k 10 900 python tools/ab_sweep.py ${WLS:-c5shard,c5} base=prmf_amd/lib/libprmf_hip.so $V=prmf_amd/lib/variants/libprmf_hip_$V.so > $OUT/ab_$V.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_$V.txt || tail -5 $OUT/ab_$V.txt
exit $rc
