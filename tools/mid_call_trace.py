#!/usr/bin/env python3
"""Developer tool: one mid-size device-resident compress call (default 16 MiB of 64 KB chunks), eight times, for rocprofv3
--kernel-trace --stats (tools/ktrace.sh): which kernels a K1w call is made of.  usage: mid_call_trace.py [MiB] [lone]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = mb << 20 if mb else 65536
base = datagen.gen("silesia", max(n, 1 << 20), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(base.size); d_src.upload(base)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(base.size, 65536))
best = 1e9
for _ in range(8):
    ctx.sync(); t0 = time.perf_counter()
    ctx.deflate_raw_async(d_src, n, 65536, 1, 1, d_c); ctx.sync()
    best = min(best, time.perf_counter() - t0)
print("%d bytes (%d chunks): best of 8 %.3f ms = %.2f GB/s, out %d" % (n, (n + 65535) >> 16, best * 1e3, n / best / 1e9, ctx.result()), flush=True)
