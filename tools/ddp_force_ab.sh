# In-call matrix: the step with and without the gradient-exchange machinery (MDT_DDP_FORCE=1: RCCL path at world size 1, where the
# collective itself moves nothing) against the number of HIP hardware queues.  Usage: gpurun -- 'bash tools/ddp_force_ab.sh [queues ...]'
set -e
run() { env "$@" timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29579 bench.py --gpus 1 --steps 6 --warmup 2 --no-selfcheck --no-cpu-baseline --no-gemm-timer --no-verify-exchange 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('   ', d['value'], d['ms_per_step'])"; }
for q in ${@:-2 4 6 7 8}; do
  echo "queues $q plain:"; run MDT_DDP_FORCE=0 GPU_MAX_HW_QUEUES=$q
  echo "queues $q forced exchange:"; run MDT_DDP_FORCE=1 GPU_MAX_HW_QUEUES=$q
done
