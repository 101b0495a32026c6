# round 5, last GPU call: the default bench line twice, with the PMC traffic of the summaries collected from this code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/bench_r5_final_a.json 2> gpurun_out/bench_r5_final_a.err; echo "rc $?"
timeout 600 python bench.py > gpurun_out/bench_r5_final.json 2> gpurun_out/bench_r5_final.err; echo "rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/bench_r5_final_a.json", "gpurun_out/bench_r5_final.json"):
    j = json.loads(open(f).read().strip().splitlines()[-1]); c = j["config"]
    print(j["value"], c["compress_GBps"], c["decompress_GBps"], j["roofline"]["traffic"], j["roofline_decode"]["frac"], j["roofline_decode"]["traffic"], c["api_decompress_GBps"], c["api_decompress_vs_bound"], j["cpu_baseline"]["value"])
PY
