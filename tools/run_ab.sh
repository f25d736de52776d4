set -e
export MDT_DDP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
MDT_GEMM_DYNAMIC=1 timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b83_dyn.log 2>&1
MDT_GEMM_DYNAMIC=0 timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b83_static.log 2>&1
MDT_GEMM_DYNAMIC=1 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/t83.log 2>&1
