# round 5, twenty-third GPU call: the timeline of a configuration that runs after another one in the same process (61.5 ms against 50.3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
API_PASSES=2 QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 4 4:6,19,44 > gpurun_out/r5w_trace.log 2>&1
grep -v "pipe\]" gpurun_out/r5w_trace.log | tail -2 | cut -c1-250; grep "pipe\]" gpurun_out/r5w_trace.log | tail -24
