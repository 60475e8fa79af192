R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
rc=0
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest_pp2.log | head; exit $rc; }
L=prmf_amd/lib/variants/libprmf_hip
SHORT=1 ROUNDS=3 timeout -k 10 1000 python tools/ab_sweep.py c5shard,c5 base=prmf_amd/lib/libprmf_hip.so pp1=${L}_pp.so pp2=${L}_pp2.so bhot16=${L}_bhot16.so pp1hot=${L}_pp1hot.so pp2hot=${L}_pp2hot.so > $OUT/ab_pp2.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_pp2.txt || tail -5 $OUT/ab_pp2.txt
exit $rc
