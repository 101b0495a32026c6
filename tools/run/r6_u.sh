# round 6: phase A with eight lanes a segment and three waves a SIMD (168 VGPRs, 96 spilled) against the default (four lanes, two waves)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6u_inflate.log
for v in "default 4" "socc3 8" "default 8" "socc3 8" "default 4"; do
  set -- $v
  if [ $1 = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$1.so; fi
  echo "== $1 K=$2" >> gpurun_out/r6u_inflate.log
  QATZIP_AMD_INFLATE_K=$2 timeout 600 python tools/inflate_var_run.py 4096:64 >> gpurun_out/r6u_inflate.log 2>&1
done
cat gpurun_out/r6u_inflate.log
