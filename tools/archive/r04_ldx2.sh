R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04pp; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export PRMF_HIP_LIB=$R/prmf_amd/lib/variants/libprmf_hip_ldenv.so
for rep in 1 2; do
WORKLOAD=${WL:-c3} timeout -k 10 600 python tools/sweep_knobs.py "LDX=64" "LDX=128" "LDX=192" "LDX=256" "LDX=320" "LDX=384" "LDX=512" "LDX=704" "LDXT=64" "LDXT=128" "LDXT=192" "LDXT=256" "LDXT=320" "LDXT=448" "LDXT=704" 2>&1 | grep -v amdgpu | awk '{print $1,$2,$4,$8,$NF-1, $(NF-1)}' 
done > $OUT/ldx_scan_${WL:-c3}.txt 2>&1
cat $OUT/ldx_scan_${WL:-c3}.txt
