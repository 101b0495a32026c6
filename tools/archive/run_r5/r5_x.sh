# round 5, twenty-fourth GPU call: copy streams with a priority of their own, helper contexts with one stream - the same
# configurations in one process and each in its own
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
echo "one process:" > gpurun_out/r5x_api.log
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 default 4 4:6,19,44 3:10,35 5:4,12,28,56 default >> gpurun_out/r5x_api.log 2>&1
echo "a process each:" >> gpurun_out/r5x_api.log
for cfg in default 4:6,19,44 4 3:10,35 5:4,12,28,56 6:3,9,20,40,68; do
API_PASSES=5 timeout 300 python tools/api_h2h.py 2047 $cfg >> gpurun_out/r5x_api.log 2>&1
done
cut -c1-250 gpurun_out/r5x_api.log
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
