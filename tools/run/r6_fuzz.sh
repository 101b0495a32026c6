# round 6: the randomized campaigns on the hardware, on the round's code (device ABI and qatzip.h), twenty minutes each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 1290 python tools/gpu_fuzz.py 1200 63001 2>&1 | tail -3) > gpurun_out/r6_fuzz_gpu.log
(timeout 1290 python tools/api_fuzz.py 1200 73001 2>&1 | tail -3) > gpurun_out/r6_fuzz_api.log
cat gpurun_out/r6_fuzz_gpu.log gpurun_out/r6_fuzz_api.log
