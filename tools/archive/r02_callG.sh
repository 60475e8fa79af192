OUT=gpurun_out/r02g; mkdir -p $OUT; export TMPDIR=/tmp
WORKLOAD=c5shard python -u tools/sweep_knobs.py "MT=1" "MT=1,WGS_PER_CU=2" "MT=1,NW=8" "NW=8" "BDB=1" "XCD=1" "XCD=2" "XCD=8" > $OUT/knobs_c5shard.txt 2>&1 && \
WORKLOAD=c2 python -u tools/sweep_knobs.py "MT=1" "WGS_PER_CU=3" "WGS_PER_CU=1" "XCD=1" > $OUT/knobs_c2.txt 2>&1 && \
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --reps 2 --workload c5shard > $OUT/prof_c5.log 2>&1
echo rc=$?; cat $OUT/knobs_c5shard.txt $OUT/knobs_c2.txt | grep -v amdgpu.ids
