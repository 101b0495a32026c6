cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_r5_final.json 2> gpurun_out/bench_r5_final.err
echo "rc $?"; tail -n 3 gpurun_out/bench_r5_final.err | cut -c1-200
timeout 400 python bench.py --gpus 2 --steps 1 --warmup 1 --mb 2048 --no-cpu > gpurun_out/bench_r5_2ranks.json 2> gpurun_out/bench_r5_2ranks.err
echo "rc $?"; tail -n 3 gpurun_out/bench_r5_2ranks.err | cut -c1-200
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_r5_final.json").read().strip().splitlines()[-1])
c = j["config"]
print({k: j[k] for k in ("value", "ms_per_step")}, {k: c.get(k) for k in ("compress_GBps", "decompress_GBps", "api_compress_GBps", "api_decompress_GBps", "api_compress_vs_bound", "api_decompress_vs_bound", "pcie_h2d_GBps", "pcie_d2h_GBps")})
print("raw_sweep", json.dumps(c.get("raw_sweep")))
print("lz4", json.dumps(c.get("lz4")))
print("roofline", json.dumps(j["roofline"])[:600])
print("roofline_decode", json.dumps(j.get("roofline_decode")))
print("cpu", json.dumps(j.get("cpu_baseline"))[:400])
try:
    j2 = json.loads(open("gpurun_out/bench_r5_2ranks.json").read().strip().splitlines()[-1])
    print(json.dumps({k: j2[k] for k in ("value", "n_gpus", "ms_per_step")}), json.dumps(j2["config"].get("one_stream"), indent=1)[:3000])
except Exception as e:
    print("no 2-rank json", e)
PY
