# round 5, twelfth GPU call: what the box gives the CPU baseline - cgroup limits, and the libz loop's summed rate by the number of
# pinned workers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
( echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset.cpus.effective: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"; echo "nproc: $(nproc)"; lscpu | grep -i "model name\|socket\|thread\|core(s)\|numa node(s)\|^cpu(s)" ; cat /proc/loadavg ) > gpurun_out/r5l_cpu.log 2>&1
python - >> gpurun_out/r5l_cpu.log 2>&1 <<'PY'
import sys, json, time
sys.path.insert(0, "tests")
import bench
cpus, model = bench.host_cpu()
print("host_cpu():", len(cpus), "cpus, first", cpus[:8], "last", cpus[-4:])
for nw in (1, 8, 16, 32, 64, 128):
    t0 = time.time()
    r = bench.cpu_baseline(16, 64, nw)
    print(nw, "workers:", json.dumps({k: r.get(k) for k in ("value", "cores", "kind", "spread")}), "port", r["port"]["value"], "wall %.1f s" % (time.time() - t0), flush=True)
PY
cat gpurun_out/r5l_cpu.log
