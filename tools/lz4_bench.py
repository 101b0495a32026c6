#!/usr/bin/env python3
"""Developer tool: 64 KB LZ4 frames (BASELINE config 4) over the bench data, compress / decompress GB/s, host call to host
return.  usage: lz4_bench.py [MiB]   (QATZIP_AMD_SO selects a variant build)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mb << 20
base = datagen.gen("silesia", min(128 << 20, n), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(n)
P = len(base) - 4099 if n > len(base) else len(base)
for off in range(0, n, P):
    d_src.upload(base[:min(P, n - off)], off)
nfr = n // 65536
d_c = ctx.alloc(n + nfr * 64 + 4096); d_b = ctx.alloc(n + 4096)
bc = bd = 1e9
for _ in range(3):
    t0 = time.perf_counter(); cl, lens = ctx.lz4_compress_frames(d_src, n, d_c, 65536); t1 = time.perf_counter()
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
    segs = np.zeros(nfr, qatzip_amd._lib.LZ4SEG_DT)
    segs["in_off"] = offs; segs["out_off"] = np.arange(nfr, dtype=np.int64) * 65536; segs["in_len"] = lens; segs["out_cap"] = 65536
    res = np.zeros(nfr, qatzip_amd._lib.LZ4RES_DT)
    t2 = time.perf_counter()
    ctx._chk(ctx.L.qzd_lz4_decompress_frames(ctx.h, d_c.ptr, d_b.ptr, segs.ctypes.data, nfr, res.ctypes.data))
    t3 = time.perf_counter()
    assert (res["status"] == 0).all()
    bc = min(bc, t1 - t0); bd = min(bd, t3 - t2)
assert ctx.crc32(d_b, n) == ctx.crc32(d_src, n)
print("%-24s lz4 %d MiB: compress %.2f GB/s  decompress %.2f GB/s  ratio %.4f  stream crc %08x"
      % (os.path.basename(os.environ.get("QATZIP_AMD_SO", "default")), mb, n / bc / 1e9, n / bd / 1e9, cl / n, ctx.crc32(d_c, cl)))
