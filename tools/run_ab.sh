set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp gpurun_lib_new.so multimodaldiscussiontransformer_amd/libmdt_hip.so
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k gemm > gpurun_out/t69.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_shapes_g1 -- python3 tools/gemm_pmc_shapes.py > gpurun_out/pmc_shapes_g1.log 2>&1
python tools/gemm_pmc_shapes.py --report gpurun_out/pmc_shapes_g1 > gpurun_out/pmc_g1.log 2>&1
bash tools/ab_libs.sh
