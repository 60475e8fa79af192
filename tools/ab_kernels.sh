# per-kernel average durations (rocprofv3 --kernel-trace) of bench.py under several library builds, one session
# usage: bash tools/ab_kernels.sh WORKLOAD name=lib.so ...
WL=$1; shift; OUT=gpurun_out/abk; mkdir -p $OUT; export TMPDIR=/tmp
for kv in "$@"; do
  name=${kv%%=*}; lib=${kv#*=}
  export PRMF_HIP_LIB=$PWD/$lib
  rocprofv3 --kernel-trace -d $OUT/$name -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --reps 3 --workload $WL > $OUT/$name.log 2>&1 || { echo "FAILED $name"; tail -3 $OUT/$name.log; exit 1; }
  db=$(find $OUT/$name -name "*_results.db" | head -1)
  echo "== $name: $(python3 -c "import json,sys; d=json.loads(open('$OUT/$name.log').read().strip().splitlines()[-1]); print('%.1f it/s %.4f ms' % (d['value'], d['ms_per_step']))")"
  python3 tools/rocpd_stats.py $db $OUT/$name.csv > /dev/null; grep -E "epilogue|sweep_kernel" $OUT/$name.csv | cut -c1-120
done
