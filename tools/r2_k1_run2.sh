#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q > gpurun_out/b_pytest.log 2>&1; tail -3 gpurun_out/b_pytest.log
for v in w16 w24 w32 w32r4; do QATZIP_AMD_SO=$R/build/var/lib_$v.so timeout 120 python tools/k1_var_run.py 1024 2>&1 | tail -1; done | tee gpurun_out/b_variants.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/b_bench.log 2>&1; grep '^{' gpurun_out/b_bench.log | tail -1 | cut -c1-900
