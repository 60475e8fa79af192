R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
E=prmf_amd/lib/variants/libprmf_hip_ldenv.so
ROUNDS=${ROUNDS:-5} timeout -k 10 1100 python tools/ab_sweep.py ${WLS:-c3} base=$E x64=$E:PRMF_SWEEP_LDX=64 x192=$E:PRMF_SWEEP_LDX=192 x448=$E:PRMF_SWEEP_LDX=448 t64=$E:PRMF_SWEEP_LDXT=64 t192=$E:PRMF_SWEEP_LDXT=192 b64=$E:PRMF_SWEEP_LDX=64:PRMF_SWEEP_LDXT=64 > $OUT/ab_ldx3.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_ldx3.txt || tail -5 $OUT/ab_ldx3.txt
exit $rc
