# round 6: the decode's host side trimmed (descriptors of a uniform call written on the device, no second copy of the results, fallback lists sized lazily)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
QATZIP_AMD_TRACE=1 timeout 300 python tools/inflate_var_run.py 4096:64 2>&1 | tail -7 > gpurun_out/r6aj.log
timeout 600 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 256:64 64:64 1024:16 1024:128 1024:512 >> gpurun_out/r6aj.log 2>&1
timeout 600 python tools/inflate_var_run.py 4096:64 1024:64 >> gpurun_out/r6aj.log 2>&1
cat gpurun_out/r6aj.log
timeout 1500 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_api.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py tests/test_gpu_cli.py -x -q 2>&1 | tail -3
