# round 5, twentieth GPU call: qzDecompress host to host, pieces that grow (a small first piece lets the output leave early;
# every later one must be decoded before the one in front of it has left) - five timed passes each, and the timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python tools/api_h2h.py 2047 default 3 4 4:6,19,44 4:8,24,52 3:10,35 5:4,12,28,56 > gpurun_out/r5t_api.log 2>&1
API_PASSES=2 QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 4:6,19,44 > gpurun_out/r5t_trace.log 2>&1
cat gpurun_out/r5t_api.log; grep "pipe\]" gpurun_out/r5t_trace.log | tail -24
