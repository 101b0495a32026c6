#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in w24 w24ng w24ns w24ngs w28; do QATZIP_AMD_SO=$R/build/var/lib_$v.so timeout 120 python tools/k1_var_run.py 1024 2>&1 | tail -2; done | tee gpurun_out/d_variants.log
