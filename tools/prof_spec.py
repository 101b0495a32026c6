#!/usr/bin/env python3
"""Developer tool: where a wave of the K-lanes-per-segment phase A spends its shader clocks (a -DQZK_SPEC_PROF build:
QATZIP_AMD_SO=build/var/lib_sprof.so).  usage: prof_spec.py [MiB] [chunk KiB]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ck = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = mb << 20
base = datagen.gen("silesia", min(128 << 20, n), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(n)
P = len(base) - 4099 if n > len(base) else len(base)
for off in range(0, n, P):
    d_src.upload(base[:min(P, n - off)], off)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(n, ck << 10)); d_o = ctx.alloc(n)
ctx.deflate_raw_async(d_src, n, ck << 10, 1, 1, d_c); ctx.sync()
clen = ctx.result()
os.environ["QATZIP_AMD_INFLATE"] = "lane"
ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=False)
assert ctx.L.qzd_spec_prof(None, C.c_uint32(0)) == 0           # reset (the atomics accumulate)
ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=False)
nseg = n // (ck << 10)
K = int(os.environ.get("QATZIP_AMD_INFLATE_K", "16" if nseg <= 8192 and ck > 16 else "8" if nseg <= (32768 if ck > 16 else 16384) else "4"))
spw = 64 // K
nw = min(8192, (nseg + spw - 1) // spw + 1)                   # + 1: a candidate that is no segment makes one more
buf = np.zeros((nw, 8), np.uint64)
assert ctx.L.qzd_spec_prof(buf.ctypes.data_as(C.c_void_p), C.c_uint32(nw)) == 0
b = buf.astype(np.float64)
look_clk = (buf[:, 3] & np.uint64((1 << 40) - 1)).astype(np.float64); look_n = (buf[:, 3] >> np.uint64(40)).astype(np.float64)
b[:, 3] = 0                                                     # (column 3: the looks at trails, part of the decode rounds' clocks)
tot = b[:, :4].sum(1)
print("%d MiB / %d KiB chunks: %d waves; shader clocks per wave mean %.2f M, max %.2f M" % (mb, ck, nw, tot.mean() / 1e6, tot.max() / 1e6))
for w, label in ((slice(None), "all waves"), (np.argsort(-tot)[:max(1, nw // 16)], "the slowest sixteenth")):
    x = b[w]; t = x[:, :4].sum()
    print("  %s (%.2f M clocks a wave):" % (label, t / len(x) / 1e6))
    for k, name in enumerate(["block headers (lane 0 of every group)", "starts, long-code tables", "decode rounds and chain walks"]):
        print("    %-40s %5.1f %%" % (name, 100 * x[:, k].sum() / t))
    lc, ln = look_clk[w], look_n[w]
    print("    of the rounds: looks at a neighbour's trail %.1f %% of the wave's clocks - %.0f of a wave's %.0f trips have a lane looking, %.0f clocks each" %
          (100 * lc.sum() / t, ln.mean(), x[:, 6].mean(), lc.sum() / max(1.0, ln.sum())))
    print("    inside the rounds: the busiest lane is in the hot loop %.1f %% of the round time; its trips %.0f a wave, %.0f clocks a trip;"
          " %.0f trips a lane on average" %
          (100 * x[:, 4].sum() / x[:, 2].sum(), x[:, 6].mean(), x[:, 4].sum() / x[:, 6].sum(), x[:, 7].sum() / 64 / len(x)))
    print("    (K = %d lanes per segment, %d segments a wave)" % (K, spw))
# waves in launch order (the host seats the largest compressed segments first): clocks by position in the launch
q = max(1, nw // 16)
print("  by place in the launch (sixteenths): mean / max M clocks a wave, mean headers %")
for i in range(0, nw, q):
    x = b[i:i + q]; t = x[:, :4].sum(1)
    print("    waves %5d..%5d: %6.2f / %6.2f   headers %4.1f %%  busiest lane trips %5.0f  rounds' clocks per trip %5.0f" %
          (i, min(nw, i + q) - 1, t.mean() / 1e6, t.max() / 1e6, 100 * x[:, 0].sum() / t.sum(), x[:, 6].mean(), x[:, 2].sum() / max(1, x[:, 6].sum())))

# the launch on the wall clock (s_memrealtime, 100 MHz): when waves began and ended, how many ran at once
beg = ((buf[:, 5] >> np.uint64(32)) & np.uint64(0xffffffff)).astype(np.int64); end = (buf[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
end = np.where(end < beg, end + (1 << 32), end)
t0 = beg.min(); beg = (beg - t0) / 100.0; end = (end - t0) / 100.0      # microseconds
stamp = np.zeros(8, np.uint64)
assert ctx.L.qzd_spec_prof(stamp.ctypes.data_as(C.c_void_p), C.c_uint32(0xffffffff)) == 0
sm = (stamp.astype(np.int64) - np.int64(stamp[0])) / 1e5            # ms since the marker scan's first wave
print("  wall clock, ms since the marker scan began: phase A's first wave enters at %.2f, its last leaves at %.2f; phase B's first wave enters at %.2f"
      % (sm[2], sm[3], sm[4]))
beg_abs = ((buf[:, 5] >> np.uint64(32)) & np.uint64(0xffffffff)).astype(np.int64); end_abs = (buf[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
s0 = int(stamp[0]) & 0xffffffff
print("  the same clock, per wave: round loops begin %.2f .. %.2f ms, waves end %.2f .. %.2f ms after the marker scan began" %
      ((beg_abs.min() - s0) / 1e5, (beg_abs.max() - s0) / 1e5, (end_abs.min() - s0) / 1e5, (end_abs.max() - s0) / 1e5))
late = np.argsort(-end_abs)[:6]
print("  the waves that end last: " + ", ".join("place %d: %.2f..%.2f ms" % (i, (beg_abs[i] - s0) / 1e5, (end_abs[i] - s0) / 1e5) for i in late))
ms = ctx.inflate_timing()
print("  HIP events of the same call: kernels %.2f ms of which phase B %.2f" % (ms[0], ms[2]))
print("  shader clocks per tick of the wall clock, mean over waves: %.2f" % (tot / np.maximum(1.0, (end - beg) * 100.0)).mean())
print("  wall clock: the launch's waves ran from 0 to %.2f ms; a wave lasts %.2f ms on average (%.2f max)" % (end.max() / 1e3, (end - beg).mean() / 1e3, (end - beg).max() / 1e3))
for lo in np.linspace(0, end.max(), 13)[:-1]:
    hi = lo + end.max() / 12
    running = ((beg < hi) & (end > lo)).sum()
    started = ((beg >= lo) & (beg < hi)).sum()
    print("    %6.2f..%6.2f ms: %5d waves running, %5d begun (launch places %s)" % (lo / 1e3, hi / 1e3, running, started,
          "%d..%d" % (np.flatnonzero((beg >= lo) & (beg < hi)).min(), np.flatnonzero((beg >= lo) & (beg < hi)).max()) if started else "-"))
