set -e
for d in 2 3 4; do
  echo "== dist $d" >> gpurun_out/kb29.log
  MDT_GEMM_PP_DIST=$d python tools/kbench.py --gemm-only >> gpurun_out/kb29.log 2>&1
done
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/t29.log 2>&1
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench29.log 2>&1
