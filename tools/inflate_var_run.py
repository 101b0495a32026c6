#!/usr/bin/env python3
"""Developer tool: one library (QATZIP_AMD_SO = a variant from tools/inflate_variants.sh, default the product build) over
the bench's data: whole-call inflate at the sizes / chunk sizes given, phase A and phase B by HIP events, the output's CRC
against the input's.  usage: inflate_var_run.py [MiB:chunkKiB ...]   (default 4096:64 1024:64 1024:128 1024:16)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

shapes = [tuple(int(x) for x in a.split(":")) for a in (sys.argv[1:] or ["4096:64", "1024:64", "1024:128", "1024:16"])]
top = max(mb for mb, _ in shapes) << 20
base = datagen.gen(os.environ.get("SWEEP_KIND", "silesia"), min(128 << 20, top), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(top)
P = len(base) - 4099 if top > len(base) else len(base)         # the bench's tiling: no two chunks of the buffer are equal
for off in range(0, top, P):
    d_src.upload(base[:min(P, top - off)], off)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(top, 16384))
d_o = ctx.alloc(top)
name = os.path.basename(os.environ.get("QATZIP_AMD_SO", "default"))
for mb, ck in shapes:
    n = mb << 20
    ctx.deflate_raw_async(d_src, n, ck << 10, 1, 1, d_c); ctx.sync()
    clen = ctx.result()
    want = ctx.crc32(d_src, n)
    best = 1e9; bms = None
    for _ in range(3):
        t0 = time.perf_counter()
        iu, ol, crc = ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=True)
        dt = time.perf_counter() - t0
        if dt < best:
            best = dt; bms = ctx.inflate_timing()
    ok = (iu, ol, crc) == (clen, n, want)
    try:
        scr = ctx.L.qzd_inflate_scratch_bytes(ctx.h) / n
    except AttributeError:                                     # a variant library older than the getter
        scr = float("nan")
    print("%-14s %5d MiB / %3d KiB chunks: inflate %6.2f GB/s  wall %7.2f ms  kernels %7.2f = A %6.2f + B %6.2f, crc %5.2f ms  scratch %.2f x output  %s"
          % (name, mb, ck, n / best / 1e9, best * 1e3, bms[0], bms[0] - bms[2], bms[2], bms[1], scr, "OK" if ok else "MISMATCH"), flush=True)
