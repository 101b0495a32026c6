# round 5, eleventh GPU call: the LZ4 compressor's first probe window (4 / 8 / 16 / 32 probes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
: > gpurun_out/r5k_lz4.log
for v in default w04 w08 w32 default; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/lz4_bench.py 1024 >> gpurun_out/r5k_lz4.log 2>&1
done
unset QATZIP_AMD_SO
cat gpurun_out/r5k_lz4.log
