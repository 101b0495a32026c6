# round 6: phase A, waves per SIMD (2: 237 VGPRs; 3: 168 VGPRs, ~95 spilled) x lanes per segment, over the call sizes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r6w_inflate.log
for v in "default 4" "default 8" "default 16" "socc3 4" "socc3 8" "socc3 16" "socc3 32"; do
  set -- $v
  if [ $1 = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$1.so; fi
  echo "== $1 K=$2" >> gpurun_out/r6w_inflate.log
  QATZIP_AMD_INFLATE_K=$2 timeout 900 python tools/inflate_var_run.py 4096:64 2048:64 1024:64 512:64 256:64 64:64 1024:16 1024:128 1024:512 >> gpurun_out/r6w_inflate.log 2>&1
done
cat gpurun_out/r6w_inflate.log
