R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
L=prmf_amd/lib/variants/libprmf_hip
ROUNDS=5 timeout -k 10 1100 python tools/ab_sweep.py c3 pp=prmf_amd/lib/libprmf_hip.so lps3=${L}_lps3.so lps5=${L}_lps5.so prio0=${L}_prio0.so > $OUT/ab_pp7.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_pp7.txt || tail -5 $OUT/ab_pp7.txt
exit $rc
