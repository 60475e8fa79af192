OUT=gpurun_out/r02l; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -q -rA -p no:cacheprovider -x > $OUT/pytest.log 2>&1; rc=$?; echo pytest_rc=$rc; tail -3 $OUT/pytest.log | cut -c1-200
if [ $rc -ne 0 ]; then echo "tests failed: no further GPU step"; exit 1; fi
python -u tools/ab_sweep.py c3,c2 unit1=prmf_amd/lib/libprmf_hip.so unit2=prmf_amd/lib/variants/libprmf_hip_unit2.so > $OUT/ab_unit.txt 2>&1 && \
python -u tools/time_rank_share.py 8 > $OUT/rank_share8_unit1.txt 2>&1 && \
PRMF_HIP_LIB=$PWD/prmf_amd/lib/variants/libprmf_hip_unit2.so python -u tools/time_rank_share.py 8 > $OUT/rank_share8_unit2.txt 2>&1
echo rc=$?; grep MEDIAN $OUT/ab_unit.txt; tail -1 $OUT/rank_share8_unit1.txt; tail -1 $OUT/rank_share8_unit2.txt
# 2-rank rehearsal of bench.py through the RCCL test double (both ranks on the one GPU; not a measurement)
g++ -O2 -std=c++17 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/native/mock_rccl.cpp -o /tmp/libmock_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread && \
PRMF_RCCL_LIB=/tmp/libmock_rccl.so PRMF_BENCH_BACKEND=gloo PRMF_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --workload c2 --reps 2 > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank_rehearsal.err
echo rehearsal_rc=$?; tail -c 1500 $OUT/bench_2rank_rehearsal.json
