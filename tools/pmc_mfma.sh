# MFMA-utilisation counters of the sweeps (north_star: "MFMA utilisation on the dense GEMMs against gfx950 peak")
#   bash tools/pmc_mfma.sh c3 c3f32 c5shard c2   -> gpurun_out/r04pmc/pmc_mfma_<workload>.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for wl in "$@"; do
  rm -rf $OUT/raw_$wl
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/raw_$wl -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --reps 1 --workload $wl > $OUT/pmc_mfma_$wl.log 2>&1 || { tail -5 $OUT/pmc_mfma_$wl.log; exit 1; }
  python3 $R/tools/pmc_mfma_summary.py $OUT/raw_$wl $OUT/pmc_mfma_$wl.json $wl
  rm -rf $OUT/raw_$wl
done
