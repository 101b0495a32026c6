#!/usr/bin/env python3
"""Developer tool: one of bench.py's extra legs alone, the way bench.py runs it, for the profiler
(tools/profile_round.sh: rocprofv3 --kernel-trace --stats and the --pmc passes of BASELINE configs 3 and 4).
usage: legs_run.py lz4|raw16|raw64|raw128 [MiB]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

leg = sys.argv[1]
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = mb << 20
base = datagen.gen("silesia", min(128 << 20, n), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(n)
tile = len(base) - bench.TILE_SKEW if n > len(base) else len(base)
for off in range(0, n, tile):
    d_src.upload(base[:min(tile, n - off)], off)
if leg == "lz4":
    print(json.dumps(bench.lz4_leg(ctx, qatzip_amd, d_src, mb)))
else:
    print(json.dumps(bench.raw_sweep(ctx, qatzip_amd, d_src, mb, (int(leg[3:]) << 10,))))
