# in-call A/B (boxes differ by up to ~10 %: only numbers from one gpurun call are comparable)
set -e
out=gpurun_out/ab31.log; rm -f $out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_dropout_gpu.py -m gpu -x -q > gpurun_out/t31.log 2>&1
for cfg in "MDT_GEMM_PERSIST=1" "MDT_GEMM_PERSIST=0" "MDT_GEMM_PERSIST=1" "MDT_GEMM_PERSIST=0"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $out
done
for cfg in "MDT_GEMM_PERSIST=1" "MDT_GEMM_PERSIST=0"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python tools/kbench.py --gemm-only 2>/dev/null | grep "fwd\|dgrad\|gelu" >> $out
done
