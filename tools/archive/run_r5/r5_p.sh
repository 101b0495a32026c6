# round 5, sixteenth GPU call: QZK_SPEC_REACH 2 / 3 / 4 / 6 / 8, and other lane counts at 1 and 4 GiB
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5p_phaseA.log
for v in reach2 reach3 reach4 reach6 default reach2 reach3 reach4; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/inflate_var_run.py 64:64 256:64 1024:64 4096:64 1024:128 1024:16 >> gpurun_out/r5p_phaseA.log 2>&1
done
for v in reach3 reach4; do for k in 4 8 16; do
  echo "K=$k" >> gpurun_out/r5p_phaseA.log
  QATZIP_AMD_INFLATE_K=$k QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so timeout 300 python tools/inflate_var_run.py 1024:64 4096:64 2048:64 >> gpurun_out/r5p_phaseA.log 2>&1
done; done
cut -c1-170 gpurun_out/r5p_phaseA.log
