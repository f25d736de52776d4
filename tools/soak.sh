# Soak: three tenants on one card for a few minutes — every launch output of 150 training steps checked for non-finite / absurd
# values, the same step's gradients compared across 150 repetitions (twice, two model sizes).  GPU box only: bash tools/soak.sh [tag]
tag=${1:-soak}
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=4
timeout -k 10 900 python tools/finite_hunt.py --reps 150 --tag hunt > gpurun_out/${tag}_hunt.log 2>&1 &
p1=$!
timeout -k 10 900 python tools/step_determinism.py --reps 150 --tag det > gpurun_out/${tag}_det.log 2>&1 &
p2=$!
timeout -k 10 900 python tools/step_determinism.py --reps 150 --trees 8 --nodes 48 --tag det2 > gpurun_out/${tag}_det2.log 2>&1 &
p3=$!
wait $p1; r1=$?
wait $p2; r2=$?
wait $p3; r3=$?
echo "exit codes $r1 $r2 $r3"
grep -h "steps clean\|FIRST BAD\|arena bad" gpurun_out/${tag}_hunt.log | tail -3
grep -h "WORST\|NON-FINITE" gpurun_out/${tag}_det.log gpurun_out/${tag}_det2.log | tail -4
