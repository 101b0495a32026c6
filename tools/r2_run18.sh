#!/bin/bash
# fused K1+K2 against the separate launches: parity first, then the bench line for both
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
for v in 1 0; do echo "== QATZIP_AMD_FUSE=$v"; QATZIP_AMD_FUSE=$v timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'], r['roofline']['other_kernels_ms'])"; done | tee gpurun_out/q_fuse.log
