cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 200 python tools/gpu_fuzz.py 110 40001 2>&1 | tail -3) > gpurun_out/fuzz_gpu.log
(timeout 200 python tools/api_fuzz.py 110 50001 2>&1 | tail -3) > gpurun_out/fuzz_api.log
cat gpurun_out/fuzz_gpu.log gpurun_out/fuzz_api.log
