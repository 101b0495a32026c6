# round 6: K1's candidate fetch without branches (ring reads of all four at once) on the round-5 kernel; against c0 (round 5) and pf (cache + asked ahead)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_deflate.py -x -q > gpurun_out/r6f_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r6f_pytest.log
tail -n 3 gpurun_out/r6f_pytest.log
: > gpurun_out/r6f_k1.log
for v in default c0 pf default c0; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6f_k1.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r6f_k1.log
timeout 300 python tools/prof_lz77.py silesia 12288 > gpurun_out/r6f_prof.log 2>&1; cat gpurun_out/r6f_prof.log
