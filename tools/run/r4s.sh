cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 200 python tools/mid_calls.py) > gpurun_out/r4s_mid.log 2>&1; tail -n 14 gpurun_out/r4s_mid.log
(bash tools/small_calls.sh) > gpurun_out/r4s_small.log 2>&1; cat gpurun_out/r4s_small.log
(timeout 600 bash tools/fleet.sh 200 8 24 48) > gpurun_out/r4s_fleet.log 2>&1; cat gpurun_out/r4s_fleet.log
timeout 400 python -m pytest tests/test_gpu_deflate.py -m gpu -x -q 2>&1 | tail -2
