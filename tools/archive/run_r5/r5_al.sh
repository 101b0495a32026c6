# round 5, last GPU call: phase A's per-segment times again (the tool summed a column that holds something else now), and the
# default bench line with the PMC traffic of the committed summaries
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONPATH=$GRAFT_REPO_ROOT
QATZIP_AMD_SO=$GRAFT_REPO_ROOT/build/var/lib_sprof.so timeout 300 python tools/prof_phaseA_tail.py 64 256 1024 4096 > gpurun_out/phaseA_tail.txt 2>&1
timeout 900 python bench.py > gpurun_out/bench_r5_final.json 2> gpurun_out/bench_r5_final.err
echo "rc $?"; tail -c 600 gpurun_out/bench_r5_final.json
