# round 5, twenty-seventh GPU call: five to seven growing pieces, and the timeline of the best so far
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 5:4,12,28,56 5:3,9,21,48 6:3,8,17,33,60 6:4,10,20,36,62 7:3,7,14,26,44,68 5:5,14,30,58 > gpurun_out/r5aa_api.log 2>&1
cut -c1-250 gpurun_out/r5aa_api.log
API_PASSES=2 QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 5:4,12,28,56 > gpurun_out/r5aa_trace.log 2>&1
grep "pipe\]" gpurun_out/r5aa_trace.log | tail -30
