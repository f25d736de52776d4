set -e
out=gpurun_out/ab50.log; rm -f $out
MDT_GEMM_W4=1 timeout -k 10 200 python tools/gemm_check.py > gpurun_out/chk50.log 2>&1
for cfg in "MDT_GEMM_W4=0" "MDT_GEMM_W4=1" "MDT_GEMM_W4=0" "MDT_GEMM_W4=1"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 200 python tools/kbench.py --gemm-only 2>/dev/null | grep "fwd\|dgrad\|gelu\|MULAUX" >> $out
done
