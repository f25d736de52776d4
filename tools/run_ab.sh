# In-call A/B template: MI355X boxes of the pool differ by up to ~10 % on the same binary, so only numbers taken
# inside ONE gpurun call are comparable.  Edit the configurations, then:  gpurun -- 'bash tools/run_ab.sh'
set -e
out=gpurun_out/ab.log; rm -f $out
for cfg in "MDT_GEMM_PERSIST=1" "MDT_GEMM_PERSIST=0" "MDT_GEMM_PERSIST=1" "MDT_GEMM_PERSIST=0"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-selfcheck 2>/dev/null | cut -c1-200 >> $out
done
