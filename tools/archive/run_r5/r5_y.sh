# round 5, twenty-fifth GPU call: timeline of the configuration that is slow in a process of its own
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
API_PASSES=2 QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 4:6,19,44 > gpurun_out/r5y_trace.log 2>&1
grep -v "pipe\]" gpurun_out/r5y_trace.log | tail -1 | cut -c1-250; grep "pipe\]" gpurun_out/r5y_trace.log | tail -24
