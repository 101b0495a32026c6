#!/usr/bin/env python3
"""Developer tool: level-1 compress of mid-size device-resident calls (64 KB chunks) by the wave-per-chunk kernel (K1,
QATZIP_AMD_K1=pull) and the workgroup-per-chunk kernel (K1w, =wide): where does one stop paying?  usage: mid_calls.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

base = datagen.gen("silesia", 96 << 20, 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(base.size); d_src.upload(base)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(base.size, 65536))
for mb in (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 96):
    n = mb << 20
    row = []
    for k in ("pull", "wide"):
        os.environ["QATZIP_AMD_K1"] = k
        best = 1e9
        for _ in range(4):
            ctx.sync(); t0 = time.perf_counter()
            ctx.deflate_raw_async(d_src, n, 65536, 1, 1, d_c); ctx.sync()
            best = min(best, time.perf_counter() - t0)
        row.append((best, ctx.result()))
    assert row[0][1] == row[1][1]
    print("%3d MiB (%4d chunks): K1 %6.2f ms %6.2f GB/s   K1w %6.2f ms %6.2f GB/s" % (mb, n >> 16, row[0][0] * 1e3, n / row[0][0] / 1e9, row[1][0] * 1e3, n / row[1][0] / 1e9), flush=True)
