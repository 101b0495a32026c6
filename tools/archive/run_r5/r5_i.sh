# round 5, ninth GPU call: phase B with the segment index in scalar registers (55 VGPRs, 8 waves per SIMD, no scratch); the
# API's decode by pieces now that phase B is lighter
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r5i_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5i_pytest.log
tail -n 3 gpurun_out/r5i_pytest.log
timeout 300 python tools/inflate_var_run.py 4096:64 1024:64 256:64 64:64 1024:128 1024:16 > gpurun_out/r5i_inflate.log 2>&1; cat gpurun_out/r5i_inflate.log
timeout 600 python tools/api_h2h.py 2047 default 2 3 4 > gpurun_out/r5i_api.log 2>&1; cat gpurun_out/r5i_api.log
