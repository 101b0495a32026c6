#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
QATZIP_AMD_SO=$R/build/var/lib_resl2.so timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_deflate.py -x -q 2>&1 | tail -3
for v in default resl2; do echo "== $v"; if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$R/build/var/lib_$v.so; fi; timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['other_kernels_ms'])"; done | tee gpurun_out/p_resl2.log
