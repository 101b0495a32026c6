# round 5, twenty-second GPU call: does the trace move the pieces, or the configurations that ran before in the same process?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5v_api.log
for cfg in default 4:6,19,44 4 4:8,24,52 5:4,12,28,56; do
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 $cfg >> gpurun_out/r5v_api.log 2>&1
done
echo "one process:" >> gpurun_out/r5v_api.log
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 4 4:6,19,44 4:6,19,44 >> gpurun_out/r5v_api.log 2>&1
cut -c1-250 gpurun_out/r5v_api.log
