set -e
out=gpurun_out/ab46.log; rm -f $out
for cfg in "MDT_X=0" "MDT_GEMM_DIAG=4" "MDT_X=0" "MDT_GEMM_DIAG=4"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python tools/kbench.py --gemm-only 2>/dev/null | grep "fwd\|dgrad" >> $out
done
