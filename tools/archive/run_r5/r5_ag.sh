# round 5, thirty-third GPU call: many threads of small calls with more hardware queues in the runtime's pool (GPU_MAX_HW_QUEUES);
# a lone segment with thirty-two lanes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
for q in 4 8 16; do
echo "GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 16
GPU_MAX_HW_QUEUES=$q timeout 60 ./build/var/bt_sweep perfmt 2 65536 2 64
done > gpurun_out/r5ag.txt 2>&1
echo "K=32, one thread" >> gpurun_out/r5ag.txt
QATZIP_AMD_INFLATE_K=32 timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 1 >> gpurun_out/r5ag.txt 2>&1
cat gpurun_out/r5ag.txt
