cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
QATZIP_AMD_TRACE=1 timeout 300 python tools/inflate_var_run.py 4096:64 > gpurun_out/r6ai.log 2>&1
cat gpurun_out/r6ai.log
