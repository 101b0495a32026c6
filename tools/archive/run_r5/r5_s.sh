# round 5, nineteenth GPU call: qzDecompress host to host after phase A's small launches got faster - pieces again, with the
# pieces' timeline (QATZIP_AMD_TRACE)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python tools/api_h2h.py 2047 default 3 4 6 3:15,50 4:10,35,65 > gpurun_out/r5s_api.log 2>&1
QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 default > gpurun_out/r5s_trace2.log 2>&1
QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 4 > gpurun_out/r5s_trace4.log 2>&1
cat gpurun_out/r5s_api.log; grep "pipe\]" gpurun_out/r5s_trace2.log | tail -16; grep "pipe\]" gpurun_out/r5s_trace4.log | tail -28
