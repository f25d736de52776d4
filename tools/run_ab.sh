# Template for an in-call A/B on the GPU box (boxes differ by several per cent: both arms must run inside ONE gpurun call).
# Usage: gpurun --timeout 900 -- 'bash tools/run_ab.sh'; edit the list of environment settings to compare.
set -e
out=gpurun_out/ab.log; rm -f $out
for cfg in "MDT_TWO_STREAMS=0" "MDT_TWO_STREAMS=1" "MDT_TWO_STREAMS=0" "MDT_TWO_STREAMS=1"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-selfcheck --no-gemm-timer 2>/dev/null | cut -c1-160 >> $out
done
