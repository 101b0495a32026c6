cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "decoded_while_it_arrives or large_host_input or sized or hold or piece" 2>&1 | tail -3
bash tools/profile_round.sh 4096 > gpurun_out/profile_round.log 2>&1
echo profiled
