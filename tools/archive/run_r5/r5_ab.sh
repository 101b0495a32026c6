# round 5, twenty-eighth GPU call: what the first piece's phase A costs when nothing runs beside it; thirty-two lanes for small launches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
API_PASSES=2 QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 2:4 > gpurun_out/r5ab_trace.log 2>&1
grep "pipe\]" gpurun_out/r5ab_trace.log | tail -12
for k in 16 32; do QATZIP_AMD_INFLATE_K=$k timeout 300 python tools/inflate_var_run.py 16:64 64:64 128:64 256:64 2>&1 | cut -c1-150; done
