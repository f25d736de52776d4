# Diagnostic (not part of the suite): two tests were seen to fail intermittently on SOME boxes of the pool — with the round's first
# library as well as with the final one, in the same call — and never on others: tests/test_dropout_gpu.py::
# test_gemm_saved_derivative_epilogue[dtype2-shape2] (8-wave persistent GEMM, K = 256, more tiles than CUs) and, inside the full
# suite only, tests/test_fp8_gpu.py::test_fp8_producer_quantised_operands_equal_stand_alone_passes.  This keeps the evidence of a
# failing run: the first failing log in full.
out=gpurun_out/flake_hunt.log; : > $out
echo "box $(cat /sys/class/drm/card*/device/unique_id 2>/dev/null | head -1)" >> $out
n=0
for i in $(seq 1 12); do
  timeout -k 10 120 python -m pytest tests/test_dropout_gpu.py -m gpu -x -q -k "saved_derivative" > gpurun_out/flake_one.log 2>&1
  tail -1 gpurun_out/flake_one.log >> $out
  if grep -q " failed" gpurun_out/flake_one.log; then n=$((n+1)); [ $n -eq 1 ] && cp gpurun_out/flake_one.log gpurun_out/flake_first_failure.log; fi
done
echo "failures: $n of 12" >> $out
timeout -k 10 300 python tools/gemm_determinism.py 200 >> $out 2>&1      # the same launches repeated: any difference is a race
cat $out | sort | uniq -c
[ -f gpurun_out/flake_first_failure.log ] && grep -n "^E \|Mismatch\|mismatch\|Greatest" gpurun_out/flake_first_failure.log | head -30
true
