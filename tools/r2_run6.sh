#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_deflate.py -x -q > gpurun_out/f_pytest.log 2>&1; tail -3 gpurun_out/f_pytest.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'], r['roofline']['other_kernels_ms'])" | tee gpurun_out/f_bench.log
