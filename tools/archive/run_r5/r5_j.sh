# round 5, tenth GPU call: qzDecompress host to host - where to cut the member (pieces and their sizes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python tools/api_h2h.py 2047 default 2:20 2:25 3:8,40 3:10,45 3:15,50 3:20,55 4:5,25,55 4:8,30,60 > gpurun_out/r5j_api.log 2>&1; cat gpurun_out/r5j_api.log
