# In-call comparison of the persistent GEMM's main loop with (a) the real operand stream, (b) no output stores,
# (c) every tile loading tile (0,0)'s panels (all fills hit L2), (d) both.  gpurun -- 'bash tools/gemm_diag2.sh'
set -e
out=gpurun_out/gemm_diag2.log; rm -f $out
for d in 0 1 8 9; do
  echo "== MDT_GEMM_DIAG=$d" >> $out
  MDT_GEMM_NO_SPEC=1 MDT_GEMM_DIAG=$d timeout -k 10 200 python tools/gemm_stamp.py 2>&1 | grep -E "^==|stamp" | awk '/^==/{print; n=0; next} {n++; last=$0} n==40{print last}' >> $out
done
cat $out
