R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06; mkdir -p $OUT; cd $R
python tools/first_step_errors.py > $OUT/first_step_errors.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/first_step_errors.txt
cd /tmp; export TMPDIR=/tmp
for WLN in c3 c5; do
  WL=$WLN MODE=colshard rocprofv3 --kernel-trace -d $OUT/share_$WLN -o t -- python3 $R/tools/time_rank_share.py 8 > $OUT/share_$WLN.log 2>&1 || { tail -5 $OUT/share_$WLN.log; exit 1; }
  db=$(find $OUT/share_$WLN -name "*_results.db" | head -1)
  python3 $R/tools/rocpd_stats.py $db $OUT/share8_${WLN}_kernel_stats.csv > $OUT/share8_${WLN}_kernel_stats.txt 2>&1
  rm -rf $OUT/share_$WLN
  echo "== $WLN share of 8"; grep -v amdgpu $OUT/share_$WLN.log | tail -2; head -14 $OUT/share8_${WLN}_kernel_stats.csv | cut -c1-160
done
