set -e
out=gpurun_out/ab62.log; rm -f $out
timeout -k 10 300 python -m pytest tests/test_dropout_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/t62.log 2>&1
for cfg in "MDT_ATTN_NO_OCC4=1" "MDT_X=1" "MDT_ATTN_NO_OCC4=1" "MDT_X=1"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-selfcheck 2>/dev/null | cut -c1-160 >> $out
done
