#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "16 2" "8 4" "8 3" "8 2"; do set -- $cfg; echo "== LPW $1 OCC $2"; QATZIP_AMD_INFLATE_LPW=$1 QATZIP_AMD_INFLATE_OCC=$2 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['other_kernels_ms'])"; done | tee gpurun_out/k_occ.log
