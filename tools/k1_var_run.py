#!/usr/bin/env python3
"""Developer tool: one K1 variant library (QATZIP_AMD_SO) over 1 GiB of the bench data: compress throughput, K1 launch
time, CRC of the stream (must agree between variants and with the oracle-checked default build)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
total = mb << 20
base = datagen.gen("silesia", min(128 << 20, total), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(total)
P = len(base) - 4099
for off in range(0, total, P):
    d_src.upload(base[:min(P, total - off)], off)
d_dst = ctx.alloc(qatzip_amd.max_deflate_len(total, 65536))
best = None
for it in range(4):
    ctx.sync(); ctx.k1_stats(reset=True); t0 = time.perf_counter()
    ctx.deflate_raw_async(d_src, total, 65536, 1, 1, d_dst); ctx.sync()
    dt = time.perf_counter() - t0
    n = ctx.result()
    k1ms, k1l, k1c = ctx.k1_stats()
    best = dt if best is None or dt < best else best
crc = ctx.crc32(d_dst, n)
ms = ctx.timing()
# parity: the streams of the first 1024 and the last 1024 chunks against the oracle (chunk streams are independent)
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
K = 1024 * 65536
lens = np.zeros(total // 65536, np.uint32)
ctx._chk(ctx.L.qzd_chunk_lens(ctx.h, lens.ctypes.data, len(lens)))
head = d_dst.download(int(lens[:1024].sum())).tobytes()
exp = O.sw_compress("RAW", d_src.download(K).tobytes(), 65536, 1, last=0, cap=K * 9 // 8 + 65536)[2]
ok1 = head == exp
tail_off = int(lens[:-1024].sum())
tail = d_dst.download(n - tail_off, tail_off).tobytes()
exp2 = O.sw_compress("RAW", d_src.download(K, total - K).tobytes(), 65536, 1, last=1, cap=K * 9 // 8 + 65536)[2]
ok2 = tail == exp2
print("parity vs oracle: first 1024 chunks %s, last 1024 chunks %s" % ("OK" if ok1 else "MISMATCH", "OK" if ok2 else "MISMATCH"))
print("%-28s compress %6.2f GB/s  K1 %.2f ms/launch (%d launches, %.0f chunks each)  first-batch K1 %.2f K2 %.2f  out %d crc %08x"
      % (os.path.basename(os.environ.get("QATZIP_AMD_SO", "default")), total / best / 1e9, k1ms / max(k1l, 1), k1l, k1c / max(k1l, 1), ms[0], ms[1], n, crc), flush=True)
