# Rehearsal of `bench.py --gpus N` on a ONE-GPU box (never used for reported numbers): N ranks share cuda:0, torch.distributed
# over gloo; once with host-driven collectives, once with the in-library step through the RCCL test double.
# usage: bash tools/rehearse_bench_multirank.sh N WORKLOAD MODE
N=${1:-2}; WL=${2:-c2}; MODE=${3:-colshard}; export TMPDIR=/tmp
MOCK=/tmp/libmock_rccl.so
g++ -O2 -std=c++17 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/native/mock_rccl.cpp -o $MOCK -L/opt/rocm/lib -lamdhip64 -lrt -lpthread || exit 1
export PRMF_BENCH_BACKEND=gloo PRMF_BENCH_SHARE_GPU=1
P=$((29500 + RANDOM % 400))
echo "== host-driven collectives (gloo), $N ranks, $WL, $MODE"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $P bench.py --gpus $N --steps 10 --warmup 3 --reps 2 --workload $WL --dist-mode $MODE 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-600
echo "== in-library step through the RCCL test double, $N ranks, $WL, $MODE"
PRMF_RCCL_LIB=$MOCK timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((P + 1)) bench.py --gpus $N --steps 10 --warmup 3 --reps 2 --workload $WL --dist-mode $MODE 2>&1 | grep -E '^\{|Error|error|Traceback' | cut -c1-600
