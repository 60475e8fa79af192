R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
L=prmf_amd/lib/variants/libprmf_hip
ROUNDS=5 timeout -k 10 1100 python tools/ab_sweep.py c4share4,c4share8 pp=prmf_amd/lib/libprmf_hip.so r4k=${L}_r4k.so r4ks=${L}_r4ks.so > $OUT/ab_pp8.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_pp8.txt || tail -5 $OUT/ab_pp8.txt
exit $rc
