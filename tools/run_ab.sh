# in-call A/B (boxes differ by up to ~10 %: only numbers from one gpurun call are comparable)
set -e
out=gpurun_out/ab34.log; rm -f $out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/t34.log 2>&1
for cfg in "MDT_X=1" "MDT_GEMM_GROUP=99" "MDT_X=1" "MDT_GEMM_GROUP=99"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $out
done
