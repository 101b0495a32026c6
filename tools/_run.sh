#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_golden.py -x -q 2>&1 | tail -2
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'])" | tee gpurun_out/slot2.log
timeout 300 python tools/prof_lz77.py silesia 12288 2>&1 | tee -a gpurun_out/slot2.log
