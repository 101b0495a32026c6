# round 5, sixth GPU call: where the eight-rank rehearsal at 256 MiB per rank stands still (per-rank trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
export QATZIP_AMD_RCCL_TIMEOUT=5 QATZIP_AMD_BENCH_LEG_TIMEOUT=100 QATZIP_AMD_BENCH_TRACE=1
timeout 600 python bench.py --gpus 8 --mb 256 --members 16 --steps 1 --no-cpu > gpurun_out/r5f_bench_8ranks.json 2> gpurun_out/r5f_bench_8ranks.err; echo "bench8/256 rc $?"
grep -o '"one_stream": {.\{0,600\}' gpurun_out/r5f_bench_8ranks.json | head -c 800; echo
grep "one-stream leg\|OneStream" gpurun_out/r5f_bench_8ranks.err | tail -60
