cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu.log 2>&1
echo "rc $?" >> gpurun_out/full_gpu.log
tail -n 15 gpurun_out/full_gpu.log
