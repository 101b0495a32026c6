# round 5, twenty-sixth GPU call: the copy streams and the helpers' streams on hardware queues of their own (CU-mask streams) -
# the same configurations in one process and each in its own
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
echo "one process:" > gpurun_out/r5z_api.log
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 default 4 4:6,19,44 5:4,12,28,56 default >> gpurun_out/r5z_api.log 2>&1
echo "a process each:" >> gpurun_out/r5z_api.log
for cfg in default 4:6,19,44 4 5:4,12,28,56; do
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 $cfg >> gpurun_out/r5z_api.log 2>&1
done
cut -c1-250 gpurun_out/r5z_api.log
