set -e
out=gpurun_out/ab42.log; rm -f $out
for cfg in "MDT_DENSE_TOKENS=0" "MDT_DENSE_TOKENS=1" "MDT_DENSE_TOKENS=0"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-220 >> $out
done
