# round 6: what a mid-size and a lone compress call are made of (kernel trace of 16 MiB and of one 64 KB chunk through K1w)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
bash tools/ktrace.sh r6ag16 tools/mid_call_trace.py 16 > gpurun_out/r6ag.log 2>&1
bash tools/ktrace.sh r6ag0 tools/mid_call_trace.py 0 >> gpurun_out/r6ag.log 2>&1
cat gpurun_out/r6ag.log
