cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_shard.py -m gpu -x -q -s 2>&1 | tail -n 25
