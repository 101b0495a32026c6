# round 6: who shares a line of K1's candidate table - the waves of G workgroups on one XCD (G = 4: sixteen waves a row; 1: a workgroup's own four)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6ab_k1.log
for v in default tabgrp1 tabgrp2 tabgrp5 default tabgrp1 tabgrp2 tabgrp5; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/k1_var_run.py 4096 >> gpurun_out/r6ab_k1.log 2>&1
done
cat gpurun_out/r6ab_k1.log
