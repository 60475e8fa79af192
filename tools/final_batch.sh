# The measurement batch behind profiles/rNN_* (run through gpurun on the final tree of a round):
#   ROUND=r06 PART=1 bash tools/final_batch.sh     default bench line (C3 + secondary + cpu baseline), kernel traces, outer loop by stage, share traces
#   ROUND=r06 PART=2 bash tools/final_batch.sh     counter passes: HBM traffic (separate FETCH_SIZE / WRITE_SIZE runs) and the clock / MFMA-busy counters,
#                                                  both stamped with the sweep sources' hash (bench.py reports them only while it matches)
# WLS="c5 c5shard" restricts the workloads.  Results under gpurun_out/$ROUND/final; copy what is to be kept into profiles/.
R=$GRAFT_REPO_ROOT; ROUND=${ROUND:-r06}; OUT=$R/gpurun_out/$ROUND/final; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
if [ "${PART:-1}" = 1 ]; then
  python -u bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "default bench failed"; tail -5 $OUT/bench_default.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]; o=d["outer_loop"]
print("C3 %.1f it/s %.4f ms | %s frac %.3f | hbm %.3f mfma %.3f | stream %.0f GB/s of-stream %.3f | traffic %s" % (d["value"], d["ms_per_step"], r["bound"], r["frac"], r["hbm"]["frac"], r["mfma"]["frac"], r.get("stream_gbs",0), r.get("frac_of_stream",0), r["traffic_ratio"]))
print("outer loop: first %.3f s, warm %.3f s (%.0f it/s), device-resident %.3f s (%.0f it/s), inner %.3f s" % (o["seconds_first_call"], o["seconds_warm_call"], o["it_per_s_warm"], o["seconds_device_resident"], o["it_per_s_device_resident"], o["seconds_inner_iterations"]))
print("cpu baseline: faithful %.3f it/s, lean %.3f, %d threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["lean_value"], d["cpu_baseline"]["cores"]))
for k,v in d.get("secondary",{}).items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a not in ("note","workload")})
PY
  for wl in ${WLS:-c2 c3f32 c5 c5shard}; do
    python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary none --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || { echo "bench $wl failed"; tail -3 $OUT/bench_$wl.err; exit 1; }
  done
  for wl in ${WLS:-c3 c2 c5 c5shard c3f32}; do
    rocprofv3 --kernel-trace --stats -d $OUT/prof_$wl -o $wl -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --reps 2 --workload $wl > $OUT/prof_$wl.log 2>&1 || { echo "trace $wl failed"; tail -5 $OUT/prof_$wl.log; exit 1; }
    python3 tools/rocpd_stats.py $(find $OUT/prof_$wl -name "*_results.db" | head -1) $OUT/${wl}_kernel_stats.csv > $OUT/${wl}_kernel_stats.txt 2>&1; rm -rf $OUT/prof_$wl
    grep -i "sweep\|epilogue" $OUT/${wl}_kernel_stats.csv | cut -c1-110
  done
  TOP=25 python tools/profile_outer_loop.py > $OUT/outer_loop_profile.txt 2>&1; grep "^call\|^warm\|^set_x" $OUT/outer_loop_profile.txt | cut -c1-400
  cd /tmp
  for WLN in c3 c5; do
    WL=$WLN MODE=colshard rocprofv3 --kernel-trace -d $OUT/share_$WLN -o t -- python3 $R/tools/time_rank_share.py 8 > $OUT/share_$WLN.log 2>&1 || { tail -5 $OUT/share_$WLN.log; exit 1; }
    python3 $R/tools/rocpd_stats.py $(find $OUT/share_$WLN -name "*_results.db" | head -1) $OUT/share8_${WLN}_kernel_stats.csv > $OUT/share8_${WLN}_kernel_stats.txt 2>&1; rm -rf $OUT/share_$WLN
  done
else
  cd /tmp
  for wl in ${WLS:-c3 c2 c5 c3f32 c5shard}; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${wl}_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --reps 1 --workload $wl > $OUT/pmc_${wl}_fetch.log 2>&1 || { tail -5 $OUT/pmc_${wl}_fetch.log; exit 1; }
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${wl}_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --reps 1 --workload $wl > $OUT/pmc_${wl}_write.log 2>&1 || { tail -5 $OUT/pmc_${wl}_write.log; exit 1; }
    python3 $R/tools/pmc_summary.py $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write $OUT/pmc_traffic_${wl}.json $wl > $OUT/pmc_traffic_${wl}.txt 2>&1
    echo "$wl traffic done"; rm -rf $OUT/pmc_${wl}_fetch $OUT/pmc_${wl}_write
    rm -rf $OUT/raw_$wl
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/raw_$wl -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --reps 1 --workload $wl > $OUT/pmc_mfma_$wl.log 2>&1 || { tail -5 $OUT/pmc_mfma_$wl.log; exit 1; }
    python3 $R/tools/pmc_mfma_summary.py $OUT/raw_$wl $OUT/pmc_mfma_$wl.json $wl
    rm -rf $OUT/raw_$wl
  done
  grep -h -i "sweep" $OUT/pmc_traffic_*.txt | cut -c1-200
fi
