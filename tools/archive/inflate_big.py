#!/usr/bin/env python3
"""Developer tool: the bench's 4 GiB as ONE raw-deflate stream (two 2 GiB compress calls, last = 0 / 1, back to back in one
buffer) inflated in one call, beside the same bytes inflated as two calls.  usage: inflate_big.py [MiB per half]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

half = (int(sys.argv[1]) if len(sys.argv) > 1 else 2048) << 20
n = 2 * half
base = datagen.gen("silesia", 128 << 20, 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(n)
P = len(base) - 4099
for off in range(0, n, P):
    d_src.upload(base[:min(P, n - off)], off)
d_c = ctx.alloc(2 * qatzip_amd.max_deflate_len(half, 65536))


def view(buf, off, k):
    v = qatzip_amd.DevBuf.__new__(qatzip_amd.DevBuf)
    v.ctx, v.nbytes, v.ptr = buf.ctx, k, buf.ptr + off
    return v


# one stream: the first half leaves its stream open (last = 0), the second closes it
ctx.deflate_raw_async(view(d_src, 0, half), half, 65536, 1, 0, d_c); ctx.sync(); c0 = ctx.result()
ctx.deflate_raw_async(view(d_src, half, half), half, 65536, 1, 1, view(d_c, c0, d_c.nbytes - c0)); ctx.sync(); c1 = ctx.result()
# two streams of their own (what two calls of the bench make)
d_a = ctx.alloc(qatzip_amd.max_deflate_len(half, 65536)); d_b = ctx.alloc(qatzip_amd.max_deflate_len(half, 65536))
ctx.deflate_raw_async(view(d_src, 0, half), half, 65536, 1, 1, d_a); ctx.sync(); a0 = ctx.result()
ctx.deflate_raw_async(view(d_src, half, half), half, 65536, 1, 1, d_b); ctx.sync(); b0 = ctx.result()
d_o = ctx.alloc(n)
want = ctx.crc32(d_src, n)
for label, calls in (("two calls", [(d_a, 0, a0, 0, half), (d_b, 0, b0, half, half)]), ("one call", [(d_c, 0, c0 + c1, 0, n)])):
    best = 1e9; kern = None
    for _ in range(3):
        t0 = time.perf_counter(); ks = []
        for (buf, co, cl, oo, ol_) in calls:
            iu, ol, crc = ctx.inflate_stream(view(buf, co, cl), cl, view(d_o, oo, ol_), 65536, want_crc=False)
            assert (iu, ol) == (cl, ol_), (iu, ol, cl, ol_)
            ks.append(ctx.inflate_timing())
        dt = time.perf_counter() - t0
        if dt < best:
            best, kern = dt, ks
    ok = ctx.crc32(d_o, n) == want
    print("%-10s %s: %6.2f GB/s  wall %.1f ms  kernels %s  resolve %s  %s" % (os.path.basename(os.environ.get("QATZIP_AMD_SO", "default")), label,
          n / best / 1e9, best * 1e3, "+".join("%.1f" % k[0] for k in kern), "+".join("%.1f" % k[2] for k in kern), "OK" if ok else "MISMATCH"), flush=True)
