#!/bin/bash
# phase A: literals after a match in the same trip
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_golden.py tests/test_gpu_api.py -x -q 2>&1 | tail -2
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'], r['roofline']['other_kernels_ms'])" | tee gpurun_out/x_trip.log
