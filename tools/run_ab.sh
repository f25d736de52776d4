set -e
cp gpurun_lib_new.so multimodaldiscussiontransformer_amd/libmdt_hip.so
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > gpurun_out/t82.log 2>&1
MDT_BENCH_GEMM_TABLE=1 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/b82.log 2> gpurun_out/b82.err
bash tools/ab_libs.sh
