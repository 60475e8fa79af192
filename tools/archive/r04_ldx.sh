R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
L=prmf_amd/lib/variants/libprmf_hip
ROUNDS=${ROUNDS:-3} timeout -k 10 1100 python tools/ab_sweep.py ${WLS:-c3} base=prmf_amd/lib/libprmf_hip.so ldx64=${L}_ldx64.so ldx192=${L}_ldx192.so ldx448=${L}_ldx448.so ldx1088=${L}_ldx1088.so > $OUT/ab_ldx.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_ldx.txt || tail -5 $OUT/ab_ldx.txt
exit $rc
