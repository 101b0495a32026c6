#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/j_pytest.log
gcc -O2 -std=gnu99 -I include tests/c/bt_sweep.c -o /tmp/bt_sweep -L qatzip_amd -lqatzip_amd -lpthread -Wl,-rpath,$R/qatzip_amd
for t in 1 4 16 64; do timeout 300 /tmp/bt_sweep perfmt 16 65536 2 $t; done 2>&1 | tee gpurun_out/j_perfmt.log
QATZIP_AMD_SYNC_COALESCE=0 timeout 300 /tmp/bt_sweep perfmt 16 65536 1 16 2>&1 | tee -a gpurun_out/j_perfmt.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/j_bench.log 2>&1; tail -c 6000 gpurun_out/j_bench.log
