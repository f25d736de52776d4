set -e
out=gpurun_out/ab_opt.log; rm -f $out
for cfg in "" "--with-optimizer" "" "--with-optimizer"; do
  echo "== [$cfg]" >> $out
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-selfcheck --no-gemm-timer $cfg 2>/dev/null | cut -c1-160 >> $out
done
