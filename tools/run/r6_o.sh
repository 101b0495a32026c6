cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6o_pytest.log 2>&1; tail -5 gpurun_out/r6o_pytest.log
timeout 900 python bench.py > gpurun_out/r6o_bench.json 2> gpurun_out/r6o_bench.err; tail -c 3000 gpurun_out/r6o_bench.json
