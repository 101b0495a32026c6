#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 python tools/api_fuzz.py 150 50000 2>&1 | tail -5 | tee gpurun_out/n_api_fuzz.log
timeout 200 python tools/gpu_fuzz.py 120 70000 2>&1 | tail -5 | tee gpurun_out/n_gpu_fuzz.log
