# in-call A/B (boxes differ by up to ~10 %: only numbers from one gpurun call are comparable)
set -e
out=gpurun_out/ab30.log; rm -f $out
for cfg in "MDT_GEMM_PP_DIST=4" "MDT_GEMM_PP_DIST=4 MDT_GEMM_GROUP=0" "MDT_GEMM_PP_DIST=2" "MDT_GEMM_PP_DIST=4"; do
  echo "== $cfg" >> $out
  env $cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $out
done
for cfg in "MDT_GEMM_PP_DIST=4" "MDT_GEMM_PP_DIST=4 MDT_GEMM_GROUP=0"; do
  echo "== $cfg" >> $out
  env $cfg python tools/kbench.py --gemm-only 2>/dev/null | grep "fwd\|dgrad" >> $out
done
