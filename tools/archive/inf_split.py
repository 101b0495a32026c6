#!/usr/bin/env python3
"""Developer tool: phase A / phase B of the two-phase inflate on the bench data (no two chunks equal): count-only
(Huffman decoding alone, nothing written) against the full decode, by segment class."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = mb << 20
ctx = qatzip_amd.Context(0)
base = datagen.gen("silesia", 128 << 20, 20250523)
d_src = ctx.alloc(n)
P = len(base) - 4099
for off in range(0, n, P):
    d_src.upload(base[:min(P, n - off)], off)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(n))
clen, _ = ctx.deflate_raw(d_src, n, 65536, 1, 1, d_c, want_crc=False)
lens = np.zeros(n // 65536, np.uint32)
ctx.L.qzd_chunk_lens(ctx.h, lens.ctypes.data, len(lens))
d_o = ctx.alloc(n)
offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
order = np.argsort(-lens.astype(np.int64), kind="stable")
print("ratio %.3f; compressed chunk size percentiles: %s" % (clen / n, np.percentile(lens, [1, 10, 50, 90, 99]).astype(int)))
for flags, name in ((1, "count-only"), (0, "decode+copy")):
    segs = [(int(offs[i]), int(i) * 65536, int(lens[i]) + 4, 65536, flags) for i in order]
    ctx.inflate_segments(d_c, d_o, segs)
    t0 = time.perf_counter(); res = ctx.inflate_segments(d_c, d_o, segs); dt = time.perf_counter() - t0
    ms = ctx.inflate_timing()
    assert (res["status"] >= 0).all() and (res["out_len"] == 65536).all()
    print("%-12s %4d MiB: %7.1f ms wall  %6.2f GB/s   kernels %.1f ms of which phase B %.1f" % (name, mb, dt * 1e3, n / dt / 1e9, ms[0], ms[2]))

if os.environ.get("QATZIP_AMD_SO", "").endswith("prof.so"):
    # profiling build: nblocks = trips of the hot loop, in_used = shader cycles / 64, per segment (launch order = `order`)
    trips = res["nblocks"].astype(np.int64); cyc = res["in_used"].astype(np.int64) * 64
    w = trips.reshape(-1, 16); cw = cyc.reshape(-1, 16)
    print("trips per segment: min %d  median %d  max %d;   per wave (max lane): median %d  max %d" %
          (trips.min(), np.median(trips), trips.max(), np.median(w.max(1)), w.max(1).max()))
    print("cycles per lane: min %.2e median %.2e max %.2e;  cycles per trip (wave max cyc / wave max trips): p10 %.0f  p50 %.0f  p90 %.0f" %
          (cyc.min(), np.median(cyc), cyc.max(), *np.percentile(cw.max(1) / np.maximum(w.max(1), 1), [10, 50, 90])))
    for lo, hi in ((0, 64), (64, 512), (512, 1024), (1024, 1536), (1536, 2048)):
        print("  waves %4d-%4d: trips(max lane) %6.0f  cycles %.2e  comp bytes %d" % (lo, hi, w[lo:hi].max(1).mean(), cw[lo:hi].max(1).mean(), lens[order][lo * 16:hi * 16].mean()))
