cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
ls /tmp/qatzip_amd_rccl_* 2>/dev/null | head -3
timeout 900 python bench.py > gpurun_out/bench_r4a.json 2> gpurun_out/bench_r4a.err
echo "rc $?"; tail -n 3 gpurun_out/bench_r4a.err | cut -c1-200
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_r4a.json").read().strip().splitlines()[-1])
c = j["config"]
print({k: j[k] for k in ("value", "ms_per_step")}, {k: c.get(k) for k in ("compress_GBps", "decompress_GBps", "api_compress_GBps", "api_decompress_GBps", "api_compress_vs_bound", "api_decompress_vs_bound", "pcie_h2d_GBps", "pcie_d2h_GBps")})
print("raw_sweep", json.dumps(c.get("raw_sweep")))
print("lz4", json.dumps(c.get("lz4")))
print("roofline", json.dumps(j["roofline"])[:600])
print("roofline_decode", json.dumps(j.get("roofline_decode")))
print("cpu", json.dumps(j.get("cpu_baseline"))[:400])
PY
