# Multi-process checks on a ONE-GPU box (fresh shell: no process here has touched the GPU before it starts children):
# two ranks on the card through bench.py (gloo), the two-rank gradient-exchange test, smoke().
set -e
MDT_BENCH_WATCHDOG=120 MDT_SINGLE_DEVICE=1 MDT_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 2 --trees 32 --no-cpu-baseline > gpurun_out/chk_gloo2.log 2>&1
MDT_RUN_MULTIPROC=1 timeout -k 10 300 python -m pytest tests/test_ddp_gpu.py -m gpu -x -q > gpurun_out/chk_ddp.log 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/chk_smoke.log 2>&1
