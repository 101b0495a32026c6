#!/bin/bash
# fused K1+K2, one launch per call: parity, bench, and the launch timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_golden.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'], r['roofline']['other_kernels_ms'])" | tee gpurun_out/r_single.log
cd /tmp; rm -rf $R/gpurun_out/r_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r_trace -- python $R/bench.py --mb 4096 --steps 1 --warmup 1 --no-cpu --no-extra > $R/gpurun_out/r_trace.log 2>&1
