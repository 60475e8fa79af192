# round 4, call 2: whole GPU suite on the reworked tree; where the sweep's workgroups run vs how long they take (C2, N=8 share);
# is the k = 128 factor's trip through the fabric what costs? (ablation build: factor window L2-hot)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c2; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout -k 10 1500 python -m pytest ${PYTEST_ARGS:-tests} -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?; tail -4 $OUT/pytest.log
[ $rc = 0 ] || { grep -n "Error\|^E " $OUT/pytest.log | head -20; exit $rc; }
grep "whole_round\|rank_local" gpurun_out/parity_measured.jsonl | tail -6 | cut -c1-400
OUT=$OUT SHAPE=5000,10000,16,50 XDTYPE=f32 timeout -k 10 300 python tools/wg_where.py > $OUT/wg_where_c2.txt 2>&1 || { tail -5 $OUT/wg_where_c2.txt; exit 1; }
cat $OUT/wg_where_c2.txt
OUT=$OUT SHAPE=6250,20000,64,300 timeout -k 10 300 python tools/wg_where.py > $OUT/wg_where_share8.txt 2>&1 || { tail -5 $OUT/wg_where_share8.txt; exit 1; }
cat $OUT/wg_where_share8.txt
SHORT=1 ROUNDS=3 timeout -k 10 900 python tools/ab_sweep.py c5,c5shard base=prmf_amd/lib/libprmf_hip.so bhot16=prmf_amd/lib/variants/libprmf_hip_bhot16.so > $OUT/ab_bhot.txt 2>&1; rc=$?
grep MEDIAN $OUT/ab_bhot.txt || tail -5 $OUT/ab_bhot.txt
exit $rc
