cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 512:64 256:64 64:64 1024:16 1024:128 1024:512 1024:256 > gpurun_out/r6x_inflate.log 2>&1
timeout 900 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 >> gpurun_out/r6x_inflate.log 2>&1
cat gpurun_out/r6x_inflate.log
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r6x_pytest.log 2>&1; tail -4 gpurun_out/r6x_pytest.log
timeout 900 python bench.py > gpurun_out/r6x_bench.json 2> gpurun_out/r6x_bench.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r6x_bench.json').read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["config"]["compress_GBps"], r["config"]["decompress_GBps"], r["roofline"]["frac"], r["roofline_decode"]["frac"], r["config"]["api_compress_GBps"], r["config"]["api_decompress_GBps"])
print(json.dumps(r["config"]["raw_sweep"])[:1500]); print(json.dumps(r["config"]["lz4"])[:600])
PY
