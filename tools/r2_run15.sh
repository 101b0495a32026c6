#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_api.py -x -q 2>&1 | tail -3
for v in w16 w16nt; do QATZIP_AMD_SO=$R/build/var/lib_$v.so timeout 120 python tools/k1_var_run.py 1024 2>&1 | tail -2; done | tee gpurun_out/o_variants.log
for v in w16 w16nt; do echo "== $v"; QATZIP_AMD_SO=$R/build/var/lib_$v.so timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'], r['roofline']['full_launch_alone_ms'])"; done | tee -a gpurun_out/o_variants.log
