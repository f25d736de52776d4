cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vendor_names -- python3 tools/vendor_names.py > gpurun_out/vendor_names.log 2>&1
