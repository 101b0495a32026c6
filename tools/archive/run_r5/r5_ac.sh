# round 5, twenty-ninth GPU call: the growing plan as the default (3 %, x 1.8), CRC per piece; against explicit cuts; other sizes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 default 6:3,8,17,33,60 7:2,5,10,19,35,62 6:2,6,14,30,58 default > gpurun_out/r5ac_api.log 2>&1
for mb in 64 256 1024; do API_PASSES=5 timeout 300 python tools/api_h2h.py $mb default 2 >> gpurun_out/r5ac_api.log 2>&1; done
cut -c1-250 gpurun_out/r5ac_api.log
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "decompress or pipe or piece or member" 2>&1 | tail -3
