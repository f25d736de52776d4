cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MDT_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-gemm-timer > gpurun_out/q_prof.log 2>&1
