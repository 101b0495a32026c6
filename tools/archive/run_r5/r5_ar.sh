# round 5: the default bench line on the round's last code, with the PMC traffic re-measured on it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 150 python bench.py > gpurun_out/bench_r5_final.json 2> gpurun_out/bench_r5_final.err; echo "rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_r5_final.json").read().strip().splitlines()[-1]); c = j["config"]
print(j["value"], c["compress_GBps"], c["decompress_GBps"], j["roofline"]["traffic"], j["roofline_decode"]["frac"], j["roofline_decode"]["traffic"], c["api_decompress_GBps"], c["api_decompress_vs_bound"], j["cpu_baseline"]["value"], c["raw_sweep"]["64K"]["inflate_traffic"], c["lz4"]["decompress_traffic"])
PY
