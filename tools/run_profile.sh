# final measurement set of a round: bench line, rocprofv3 kernel stats of the same command, HBM-side traffic (PMC, separate passes)
# the profiler passes run on ONE HIP stream (MDT_TWO_STREAMS=0), like bench.py's own roofline pass: with the two branches
# overlapped a kernel's recorded duration includes its wait for compute units held by the other branch
# a second kernel trace with both streams is kept for the overlap itself (${tag}_prof2)
set -e
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 8 --warmup 3 > gpurun_out/${tag}_bench.log 2>&1
MDT_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck > gpurun_out/${tag}_prof.log 2>&1
MDT_TWO_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-gemm-timer > gpurun_out/${tag}_pmc_fetch.log 2>&1
MDT_TWO_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-gemm-timer > gpurun_out/${tag}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-gemm-timer > gpurun_out/${tag}_prof2.log 2>&1
