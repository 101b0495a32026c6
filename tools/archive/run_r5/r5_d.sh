# round 5, fourth GPU call: counters of the two rebuilt LZ4 kernels (what bounds them now), the eight-rank rehearsal again
# with the leg reporting where it stands, the scratch-ceiling test.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q -k "scratch or oracle_streams" > gpurun_out/r5d_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5d_pytest.log
tail -n 3 gpurun_out/r5d_pytest.log
export QATZIP_AMD_RCCL_TIMEOUT=5 QATZIP_AMD_BENCH_LEG_TIMEOUT=200
timeout 600 python bench.py --gpus 8 --mb 64 --members 16 --steps 1 --no-cpu > gpurun_out/r5d_bench_8ranks_64.json 2> gpurun_out/r5d_bench_8ranks_64.err; echo "bench8/64 rc $?"
grep -o '"one_stream": {.\{0,900\}' gpurun_out/r5d_bench_8ranks_64.json | head -c 1200; echo
timeout 600 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --no-cpu > gpurun_out/r5d_bench_8ranks.json 2> gpurun_out/r5d_bench_8ranks.err; echo "bench8/256 rc $?"
grep -o '"one_stream": {.\{0,900\}' gpurun_out/r5d_bench_8ranks.json | head -c 1200; echo
unset QATZIP_AMD_RCCL_TIMEOUT QATZIP_AMD_BENCH_LEG_TIMEOUT
bash tools/pmc_any.sh l4 tools/legs_run.py lz4 1024
cd $GRAFT_REPO_ROOT; python tools/pmc_show.py l4 qzk_lz4 > gpurun_out/r5d_lz4_counters.txt 2>&1; cat gpurun_out/r5d_lz4_counters.txt
