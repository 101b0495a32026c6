#!/bin/bash
# waves per CU with K2 fused: 16 (default) / 16 with the small slot table / 20 / 24
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
QATZIP_AMD_SO=$R/build/var/lib_w24.so timeout 600 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_golden.py -x -q 2>&1 | tail -2
for v in default w16s w20 w24; do echo "== $v"; if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$R/build/var/lib_$v.so; fi; timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'])"; done | tee gpurun_out/w_waves.log
