cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
QATZIP_AMD_TRACE=1 timeout 90 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 1024:128 1024:16 64:64 2>&1 | grep -v "qzd_inflate_stream" | sort | uniq -c | sort -rn
timeout 300 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
