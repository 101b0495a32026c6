# round 5, seventh GPU call: the IPC transport with one slot per rank (no allocation above 2 GiB) - its tests, the eight-rank
# rehearsal at 256 MiB per rank; 512 KB segments with sixteen lanes and the longer piece list
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q > gpurun_out/r5g_shard.log 2>&1; echo "rc $?" >> gpurun_out/r5g_shard.log
tail -n 4 gpurun_out/r5g_shard.log
export QATZIP_AMD_RCCL_TIMEOUT=5 QATZIP_AMD_BENCH_LEG_TIMEOUT=200 QATZIP_AMD_BENCH_TRACE=1
timeout 600 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --no-cpu > gpurun_out/r5g_bench_8ranks.json 2> gpurun_out/r5g_bench_8ranks.err; echo "bench8/256 rc $?"
grep -o '"one_stream": {.\{0,1100\}' gpurun_out/r5g_bench_8ranks.json | head -c 1300; echo
grep "one-stream leg\|OneStream" gpurun_out/r5g_bench_8ranks.err | tail -12
unset QATZIP_AMD_RCCL_TIMEOUT QATZIP_AMD_BENCH_LEG_TIMEOUT QATZIP_AMD_BENCH_TRACE
timeout 600 python tools/inflate_var_run.py 1024:512 1024:256 128:512 4096:64 256:64 64:64 > gpurun_out/r5g_inflate.log 2>&1; cat gpurun_out/r5g_inflate.log
timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r5g_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5g_pytest.log
tail -n 3 gpurun_out/r5g_pytest.log
