set -e
out=gpurun_out/ab72.log; rm -f $out
for cfg in "GPU_MAX_HW_QUEUES=4 MDT_TWO_STREAMS=0" "GPU_MAX_HW_QUEUES=8 MDT_TWO_STREAMS=0" "GPU_MAX_HW_QUEUES=4 MDT_TWO_STREAMS=1" "GPU_MAX_HW_QUEUES=8 MDT_TWO_STREAMS=1"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-gemm-timer --no-selfcheck 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['host_issue_ms_per_step'])" >> $out
done
