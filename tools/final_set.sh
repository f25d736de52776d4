set -e
bash tools/run_profile.sh r3v6
bash tools/sq_counters.sh r3v6 || true
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dtype fp8 > gpurun_out/r3v6_bench_fp8.log 2>&1 || true
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3v6_bench_bf16_same_call.log 2>&1 || true
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --config large > gpurun_out/r3v6_bench_large.log 2>&1 || true
tail -c 400 gpurun_out/r3v6_bench.log
