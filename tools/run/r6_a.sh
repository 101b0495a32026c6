# round 6, first GPU call: K1 with the LDS entry cache - parity on the hardware, then its launch time against the round-5
# kernel (c0) and the variants (entries / waves per CU / slot tables) in ONE session, the phase profile, a bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_golden.py -x -q > gpurun_out/r6a_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r6a_pytest.log
tail -n 3 gpurun_out/r6a_pytest.log
: > gpurun_out/r6a_k1.log
for v in default c0 c0s c7 c8w12s c9w12 c9w10 default c0; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6a_k1.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r6a_k1.log
timeout 300 python tools/prof_lz77.py silesia 12288 > gpurun_out/r6a_prof.log 2>&1; cat gpurun_out/r6a_prof.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q > gpurun_out/r6a_pytest_full.log 2>&1; echo "rc $?" >> gpurun_out/r6a_pytest_full.log
tail -n 3 gpurun_out/r6a_pytest_full.log
timeout 600 python bench.py > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err; tail -c 1500 gpurun_out/r6a_bench.json
