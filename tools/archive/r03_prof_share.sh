# kernel-level profile of one rank's share (N = 8) in both multi-rank modes (host-driven phase sequence, stand-in copies)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03ps; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for M in replicated colshard; do
  MODE=$M rocprofv3 --kernel-trace -d $OUT/$M -o t -- python3 $R/tools/time_rank_share.py 8 > $OUT/$M.log 2>&1 || { tail -5 $OUT/$M.log; exit 1; }
  db=$(find $OUT/$M -name "*_results.db" | head -1)
  python3 $R/tools/rocpd_stats.py $db $OUT/share8_$M.csv > $OUT/share8_$M.txt 2>&1
  echo "== $M"; head -16 $OUT/share8_$M.csv | cut -c1-150
done
