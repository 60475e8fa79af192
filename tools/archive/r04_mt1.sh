R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
L=prmf_amd/lib/variants/libprmf_hip
for v in mt1b; do
  PRMF_HIP_LIB=$R/${L}_$v.so timeout -k 10 600 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_fullsize as t
for wl in ('c4share8','c4share4'):
    t.test_one_iteration_vs_fp64_at_full_size(wl,'bf16'); print('parity ok', wl)
" 2>&1 | grep -v amdgpu | tail -4 || exit 1
done
ROUNDS=5 timeout -k 10 1100 python tools/ab_sweep.py c4share8,c4share4 base=prmf_amd/lib/libprmf_hip.so mt1a=${L}_mt1a.so mt1b=${L}_mt1b.so > $OUT/ab_mt1.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_mt1.txt || tail -5 $OUT/ab_mt1.txt
exit $rc
