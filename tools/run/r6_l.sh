cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6l_k1.log
for v in default w10o2a w10o2b w8o2 default w10o2a w10o2b; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6l_k1.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r6l_k1.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6l_pytest.log 2>&1; tail -5 gpurun_out/r6l_pytest.log
