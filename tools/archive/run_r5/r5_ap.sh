# round 5, the campaigns on the hardware once more on the final code (the CRC-32 step and the marker scan changed after the first run)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
(QATZIP_AMD_MARKER_CHECK=1 timeout 220 python tools/gpu_fuzz.py 150 80001 2>&1 | tail -3) > gpurun_out/fuzz_gpu2.log
(timeout 220 python tools/api_fuzz.py 150 90001 2>&1 | tail -3) > gpurun_out/fuzz_api2.log
cat gpurun_out/fuzz_gpu2.log gpurun_out/fuzz_api2.log
