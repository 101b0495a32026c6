#!/usr/bin/env python3
"""Developer tool: ONE device-layer call per direction over the bench's 4 GiB (the device ABI takes 64-bit lengths): compress,
decompress, round-trip CRC, sampled chunk streams against the oracle.  usage: big_call.py [MiB]"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402
import qatzip_amd  # noqa: E402

n = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096) << 20
base = datagen.gen("silesia", 128 << 20, 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(n)
P = len(base) - 4099
for off in range(0, n, P):
    d_src.upload(base[:min(P, n - off)], off)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(n, 65536)); d_o = ctx.alloc(n)
for it in range(3):
    ctx.sync(); t0 = time.perf_counter()
    ctx.deflate_raw_async(d_src, n, 65536, 1, 1, d_c); ctx.sync()
    tc = time.perf_counter() - t0
    cl = ctx.result()
    t0 = time.perf_counter()
    iu, ol, crc = ctx.inflate_stream(d_c, cl, d_o, 65536, want_crc=True)
    td = time.perf_counter() - t0
    print("pass %d: compress %.2f GB/s (%.1f ms)  decompress %.2f GB/s (%.1f ms)  both %.2f GB/s  ratio %.4f" %
          (it, n / tc / 1e9, tc * 1e3, n / td / 1e9, td * 1e3, 2 * n / (tc + td) / 1e9, cl / n), flush=True)
assert (iu, ol) == (cl, n) and crc == ctx.crc32(d_src, n), "round trip differs"
nch = n // 65536
lens = np.zeros(nch, np.uint32)
ctx._chk(ctx.L.qzd_chunk_lens(ctx.h, lens.ctypes.data, nch))
offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
assert int(offs[-1]) == cl
rng = np.random.Generator(np.random.PCG64(5))
for k in sorted(set([0, nch - 1, nch // 2, 32767, 32768]) | set(int(x) for x in rng.integers(0, nch, 200))):
    plain = d_src.download(65536, k * 65536).tobytes()
    exp = O.sw_compress("RAW", plain, 65536, 1, last=1 if k == nch - 1 else 0, cap=80000)[2]
    assert d_c.download(int(lens[k]), int(offs[k])).tobytes() == exp, ("chunk", k)
print("round trip CRC equal, 205 sampled chunk streams byte-identical to the oracle's")
