#!/bin/bash
# match extension out of the ring + raised priority for the tree build: parity, bench A/B, profile
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_deflate.py tests/test_gpu_golden.py -x -q 2>&1 | tail -2
for v in default noprio; do echo "== $v"; if [ $v = default ]; then unset QATZIP_AMD_SO; else export QATZIP_AMD_SO=$R/build/var/lib_$v.so; fi; timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra 2>&1 | grep '^{' | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print(r['value'], r['config']['compress_GBps'], r['config']['decompress_GBps'], r['roofline']['launch_ms'])"; done | tee gpurun_out/u_ring.log
unset QATZIP_AMD_SO
timeout 300 python tools/prof_lz77.py silesia 12288 2>&1 | tee -a gpurun_out/u_ring.log
