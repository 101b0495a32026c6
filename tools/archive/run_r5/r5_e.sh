# round 5, fifth GPU call: the LZ4 block compressor after its instruction diet (by register budget), the eight-rank rehearsal
# at 256 MiB per rank with the preparation agreed upon.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_lz4.py -x -q > gpurun_out/r5e_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r5e_pytest.log
tail -n 3 gpurun_out/r5e_pytest.log
: > gpurun_out/r5e_lz4.log
for v in default l4o7 l4o8 default l4o8; do
  if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_$v.so; fi
  timeout 300 python tools/legs_run.py lz4 1024 >> gpurun_out/r5e_lz4.log 2>&1
done
unset QATZIP_AMD_SO
cut -c1-330 gpurun_out/r5e_lz4.log
export QATZIP_AMD_RCCL_TIMEOUT=5 QATZIP_AMD_BENCH_LEG_TIMEOUT=300
timeout 900 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --no-cpu > gpurun_out/r5e_bench_8ranks.json 2> gpurun_out/r5e_bench_8ranks.err; echo "bench8/256 rc $?"
grep -o '"one_stream": {.\{0,1100\}' gpurun_out/r5e_bench_8ranks.json | head -c 1400; echo
rocm-smi --showmeminfo vram 2>/dev/null | head -8
