# round 4, call 3: the cost of a hand-off in the sweep's tail (probe build) at C3, C2 and the N = 8 share
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c3; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for lib in tailprobe2; do for wl in c2 c4share8 c3; do
  PRMF_HIP_LIB=$R/prmf_amd/lib/variants/libprmf_hip_$lib.so WORKLOAD=$wl timeout -k 10 300 python tools/tail_probe.py >> $OUT/tail_probe2.txt 2>&1 || { tail -5 $OUT/tail_probe2.txt; exit 1; }
done; done
grep -v amdgpu.ids $OUT/tail_probe2.txt
