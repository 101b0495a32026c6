# round 5: the eight-rank rehearsal and the fleet shape (200 passes: start-up amortized) on the final code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
QATZIP_AMD_RCCL_TIMEOUT=5 timeout 400 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --warmup 1 --no-cpu > gpurun_out/bench_r5_8ranks.json 2> gpurun_out/bench_r5_8ranks.err
echo "rc $?"; tail -c 1500 gpurun_out/bench_r5_8ranks.json; tail -3 gpurun_out/bench_r5_8ranks.err | cut -c1-200
echo "spinning:" > gpurun_out/r5am_fleet.txt
timeout 300 bash tools/fleet.sh 200 8 48 >> gpurun_out/r5am_fleet.txt 2>&1
echo "QATZIP_AMD_SYNC=block:" >> gpurun_out/r5am_fleet.txt
QATZIP_AMD_SYNC=block timeout 300 bash tools/fleet.sh 200 8 48 >> gpurun_out/r5am_fleet.txt 2>&1
cat gpurun_out/r5am_fleet.txt | cut -c1-250
