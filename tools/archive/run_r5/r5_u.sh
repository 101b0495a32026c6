# round 5, twenty-first GPU call: the pieces' timeline with a trace that does not move them (buffered, printed after the call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
for cfg in default 4:6,19,44 4; do
API_PASSES=3 QATZIP_AMD_TRACE=1 timeout 300 python tools/api_h2h.py 2047 $cfg > gpurun_out/r5u_trace.log 2>&1
grep -v "pipe\]" gpurun_out/r5u_trace.log | tail -1 | cut -c1-250; grep "pipe\]" gpurun_out/r5u_trace.log | tail -30 | sort -k7 -n
done
