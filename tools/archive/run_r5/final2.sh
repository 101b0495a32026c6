# round 5, after the marker scan and the CRC-32 step were rebuilt: the decode tests first (stop there if they fail), then the
# profile passes, the decode counters, the bench lines and the whole GPU suite again - the summaries are bound to the sources
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu > gpurun_out/f2_inflate_tests.log 2>&1
tail -n 3 gpurun_out/f2_inflate_tests.log
grep -q " passed" gpurun_out/f2_inflate_tests.log && ! grep -q "failed\|error" gpurun_out/f2_inflate_tests.log || { echo "decode tests failed: stopping"; exit 1; }
QATZIP_AMD_TRACE=1 timeout 200 python tools/inflate_var_run.py 4096:64 2>&1 | tail -7 | cut -c1-200 > gpurun_out/f2_trace.txt; cat gpurun_out/f2_trace.txt
bash tools/profile_round.sh 4096 > gpurun_out/profile_round.log 2>&1
bash tools/run/pmc_decode.sh > gpurun_out/pmc_decode.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_show.py dec qzk_inflate_spec qzk_lz_resolve > gpurun_out/decode_counters.txt 2>&1
bash tools/run/bench_final.sh > gpurun_out/bench_final.log 2>&1
bash tools/run/full_gpu.sh > gpurun_out/full_gpu_tail.log 2>&1
tail -n 5 gpurun_out/full_gpu_tail.log; tail -n 30 gpurun_out/bench_final.log | cut -c1-700 | head -12
