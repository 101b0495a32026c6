#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py -x -q > gpurun_out/c_pytest.log 2>&1; tail -3 gpurun_out/c_pytest.log
for v in w16 w16plain w24 w24s128 w16r8 w20r4; do QATZIP_AMD_SO=$R/build/var/lib_$v.so timeout 120 python tools/k1_var_run.py 1024 2>&1 | tail -2; done | tee gpurun_out/c_variants.log
