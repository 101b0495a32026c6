cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
QATZIP_AMD_TRACE=1 timeout 120 python tools/inflate_var_run.py 4096:64 1024:64 1024:128 1024:16 256:64 64:64 16:64 1:64 64:128 128:512 16:16 2>&1 | grep -v "qzd_inflate_stream" | grep -v "^\[two_phase\].* 0 of"
timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -3
