set -e
out=gpurun_out/ab66.log; rm -f $out
export MDT_DDP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for cfg in "MDT_TWO_STREAMS=1" "MDT_TWO_STREAMS=1 GPU_MAX_HW_QUEUES=8" "MDT_TWO_STREAMS=1 GPU_MAX_HW_QUEUES=16" "MDT_TWO_STREAMS=0 GPU_MAX_HW_QUEUES=8" "MDT_TWO_STREAMS=1 MDT_DDP_FORCE=0" "MDT_TWO_STREAMS=1 MDT_DDP_FORCE=0 GPU_MAX_HW_QUEUES=8" "MDT_TWO_STREAMS=1 GPU_MAX_HW_QUEUES=8"; do
  echo "== $cfg" >> $out
  env $cfg timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-selfcheck --no-gemm-timer 2>/dev/null | cut -c1-160 >> $out
done
