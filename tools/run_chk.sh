set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t64.log 2>&1
MDT_DDP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b64_force.log 2>&1
MDT_SINGLE_DEVICE=1 MDT_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 2 --trees 32 --no-cpu-baseline > gpurun_out/b64_gloo2.log 2>&1
