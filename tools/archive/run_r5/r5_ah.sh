# round 5, thirty-fourth GPU call: a lone segment through the K-lane kernels instead of one wave
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
for w in 0 1; do
if [ $w = 1 ]; then export QATZIP_AMD_LONE_WAVE=1; echo "one wave"; else echo "lanes"; fi
timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 1
timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 16
timeout 60 ./build/var/bt_sweep perfmt 4 16384 2 1
timeout 60 ./build/var/bt_sweep perfmt 16 524288 2 1
done > gpurun_out/r5ah.txt 2>&1
unset QATZIP_AMD_LONE_WAVE
QATZIP_AMD_TRACE=1 timeout 60 ./build/var/bt_sweep perfmt 1 65536 1 1 2>&1 | tail -8 >> gpurun_out/r5ah.txt
cat gpurun_out/r5ah.txt
