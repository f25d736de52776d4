set -e
bash tools/run_profile.sh ${1:-r3v7}
bash tools/sq_counters.sh ${1:-r3v7} || true
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --dtype fp8 > gpurun_out/${1:-r3v7}_bench_fp8.log 2>&1 || true
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${1:-r3v7}_bench_bf16_same_call.log 2>&1 || true
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --config large > gpurun_out/${1:-r3v7}_bench_large.log 2>&1 || true
tail -c 400 gpurun_out/${1:-r3v7}_bench.log
