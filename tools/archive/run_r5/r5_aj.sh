# round 5, thirty-sixth GPU call: the fleet shape with waiting threads asleep (QATZIP_AMD_SYNC=block) against spinning ones
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
: > gpurun_out/r5aj_fleet.txt
export QATZIP_AMD_SYNC=block
for P in 8 16 48; do
  a=$(grep -E "throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  timeout 200 bash tools/fleet.sh 40 $P >> gpurun_out/r5aj_fleet.txt 2>&1
  b=$(grep -E "throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  echo "  cgroup cpu.stat before: $a after: $b" >> gpurun_out/r5aj_fleet.txt
done
timeout 60 ./build/var/bt_sweep perfmt 4 65536 2 1 >> gpurun_out/r5aj_fleet.txt 2>&1
timeout 60 ./build/var/bt_sweep perfmt 2 65536 2 64 >> gpurun_out/r5aj_fleet.txt 2>&1
cat gpurun_out/r5aj_fleet.txt | cut -c1-250
