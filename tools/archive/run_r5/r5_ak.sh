# round 5, thirty-seventh GPU call: a piece's CRC-32 taken while it leaves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 default 2>&1 | cut -c1-250
API_PASSES=5 timeout 300 python tools/api_h2h.py 1024 default 2>&1 | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "decoded_while_it_arrives" 2>&1 | tail -3
