# round 5: the host-side steps of a 4 GiB decode (QATZIP_AMD_TRACE), to see what is not kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
QATZIP_AMD_TRACE=1 timeout 200 python tools/inflate_var_run.py 4096:64 2>&1 | tail -12 | cut -c1-200 > gpurun_out/r5an_trace.txt
cat gpurun_out/r5an_trace.txt
