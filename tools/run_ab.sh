set -e
out=gpurun_out/ab38.log; rm -f $out
python tools/gemm_check.py > gpurun_out/chk38.log 2>&1
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/t38.log 2>&1
for i in 1 2; do timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-160 >> $out; done
timeout -k 10 200 python tools/kbench.py --gemm-only 2>/dev/null | grep "fwd\|dgrad\|gelu" >> $out
