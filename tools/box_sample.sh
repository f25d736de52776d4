# One sample of the device this call landed on: the two tests that were device-dependent in rounds 3-4 (the 8-wave GEMM's saved-derivative
# launch, root-caused in round 4; the fp8 producer route, still open), each a few times, and the op trace.  Appends one line per device to
# gpurun_out/box_samples.log.  GPU box only: bash tools/box_sample.sh
id=$(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)
f8=0; sd=0
for i in 1 2 3 4; do
  timeout -k 10 200 python -m pytest tests/test_fp8_gpu.py -q -x -k "producer_quantised" > gpurun_out/box_one.log 2>&1 || f8=$((f8+1))
  grep -q "xfailed" gpurun_out/box_one.log && f8=$((f8+1))
  timeout -k 10 120 python -m pytest tests/test_dropout_gpu.py -q -x -k "saved_derivative" > gpurun_out/box_two.log 2>&1 || sd=$((sd+1))
done
tr=$(timeout -k 10 300 python tools/op_trace.py 4 2>&1 | grep "^repetition [1-9]" | sed 's/.*first scales differing [0-9]*; //' | sort | uniq -c | tr '\n' ';')
echo "device $id: fp8 route test not green $f8 of 4; saved-derivative test failed $sd of 4; op trace: $tr" | tee -a gpurun_out/box_samples.log
