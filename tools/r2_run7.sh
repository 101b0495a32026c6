#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for lpw in 8 16 32; do echo "== LPW $lpw"; QATZIP_AMD_INFLATE_LPW=$lpw timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['other_kernels_ms'])"; done | tee gpurun_out/g_lpw.log
