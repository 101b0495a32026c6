#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
QATZIP_AMD_SO=$R/build/var/lib_prof.so timeout 300 python tools/inf_split.py 2048 2>&1 | tail -12 | tee gpurun_out/l_prof.log
