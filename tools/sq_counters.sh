# SQ counter passes over one bench step (one HIP stream, so a kernel's counters are its own): where the wave cycles of
# each kernel family go.  Usage on the GPU box: bash tools/sq_counters.sh <tag>; then python tools/sq_counters.py <tag>
set -e
tag=${1:-sq}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-gemm-timer"
MDT_TWO_STREAMS=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/${tag}_sq_a -- $B > gpurun_out/${tag}_sq_a.log 2>&1
MDT_TWO_STREAMS=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/${tag}_sq_b -- $B > gpurun_out/${tag}_sq_b.log 2>&1
