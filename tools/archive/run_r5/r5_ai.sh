# round 5, thirty-fifth GPU call: what one process holds on the device, and the fleet shape with the host's CPU quota beside it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/qatzip_amd
timeout 120 python tools/mem_probe.py > gpurun_out/r5ai_mem.txt 2>&1
cat gpurun_out/r5ai_mem.txt
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc $(nproc)" > gpurun_out/r5ai_fleet.txt
for P in 8 16 48; do
  a=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  timeout 200 bash tools/fleet.sh 40 $P >> gpurun_out/r5ai_fleet.txt 2>&1
  b=$(grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
  echo "  cgroup cpu.stat before: $a after: $b" >> gpurun_out/r5ai_fleet.txt
done
cat gpurun_out/r5ai_fleet.txt | cut -c1-250
