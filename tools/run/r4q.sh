cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 400 python bench.py --gpus 2 --steps 1 --warmup 1 --mb 2048 --no-cpu > gpurun_out/r4q_bench2.json 2> gpurun_out/r4q_bench2.err
echo "rc $?"; tail -n 5 gpurun_out/r4q_bench2.err | cut -c1-300
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/r4q_bench2.json").read().strip().splitlines()[-1])
    print(json.dumps({k: j[k] for k in ("value", "n_gpus", "ms_per_step")}), json.dumps(j["config"].get("one_stream"), indent=1))
except Exception as e:
    print("no json", e)
PY
