# round 5, thirty-first GPU call: small synchronous calls after the round's changes, and where a lone 64 KB call spends its time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
bash tools/small_calls.sh > gpurun_out/r5ae_small.txt 2>&1
QATZIP_AMD_TRACE=1 timeout 60 ./build/var/bt_sweep perfmt 1 65536 1 1 > gpurun_out/r5ae_trace.txt 2>&1
cat gpurun_out/r5ae_small.txt; tail -40 gpurun_out/r5ae_trace.txt | cut -c1-200
timeout 300 python tools/inflate_var_run.py 1024:512 1024:256 2>&1 | cut -c1-150
