cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 150 python tools/gpu_fuzz.py 70 60001 2>&1 | tail -3) > gpurun_out/fuzz_gpu.log
(timeout 150 python tools/api_fuzz.py 70 70001 2>&1 | tail -3) > gpurun_out/fuzz_api.log
cat gpurun_out/fuzz_gpu.log gpurun_out/fuzz_api.log
