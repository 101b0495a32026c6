cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python tools/api_h2h.py 2047 default 2>&1 | grep -v Warning | grep "pieces"
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_inflate.py tests/test_gpu_cli.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -5
