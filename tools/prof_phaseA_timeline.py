#!/usr/bin/env python3
"""Developer tool (round 6): phase A's waves on the wall clock (a -DQZK_SPEC_PROF build: QATZIP_AMD_SO=build/var/lib_sprof.so;
qzk_spec_prof[w][5] = begin << 32 | end on the 100 MHz clock every wave agrees on): how many waves run at each moment of the
launch, and how well the order the host launches the segments in - longest compressed length first, qzd_inflate.hip
inflate_stream - predicts a wave's duration.  usage: prof_phaseA_timeline.py [MiB]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ck = 64
n = mb << 20
base = datagen.gen("silesia", min(128 << 20, n), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(n)
P = len(base) - 4099 if n > len(base) else len(base)
for off in range(0, n, P):
    d_src.upload(base[:min(P, n - off)], off)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(n, 16384)); d_o = ctx.alloc(n)
ctx.deflate_raw_async(d_src, n, ck << 10, 1, 1, d_c); ctx.sync()
clen = ctx.result()
nseg = n // (ck << 10)
lens = np.zeros(nseg, np.uint32)
ctx._chk(ctx.L.qzd_chunk_lens(ctx.h, lens.ctypes.data, nseg))
ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=False)
assert ctx.L.qzd_spec_prof(None, C.c_uint32(0)) == 0
ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=False)
ms = ctx.inflate_timing()
K = int(os.environ.get("QATZIP_AMD_INFLATE_K", "0")) or (16 if nseg <= 32768 else 8)
spw = 64 // K
nw = min(8192, (nseg + spw - 1) // spw)
buf = np.zeros((nw, 8), np.uint64)
assert ctx.L.qzd_spec_prof(buf.ctypes.data_as(C.c_void_p), C.c_uint32(nw)) == 0
beg = (buf[:, 5] >> np.uint64(32)).astype(np.int64) & 0xffffffff
end = (buf[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
end = np.where(end < beg, end + (1 << 32), end)
t0 = beg.min()
b = (beg - t0) / 1e5; e = (end - t0) / 1e5                       # ms
dur = e - b
print("%d MiB: %d waves of %d segments, phase A %.2f ms by HIP events; first start 0, last end %.2f ms" % (mb, nw, spw, ms[0] - ms[2], e.max()))
print("  wave duration ms: p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f; sum / 2048 resident = %.2f ms" %
      (*[float(np.percentile(dur, p)) for p in (10, 50, 90, 99)], dur.max(), dur.sum() / 2048))
edges = np.arange(0, e.max() + 1, 1.0)
print("  waves running at t (ms): " + "  ".join("%d:%d" % (t, int(((b <= t) & (e > t)).sum())) for t in edges))
print("  starts per ms:           " + "  ".join("%d:%d" % (t, int(((b >= t) & (b < t + 1)).sum())) for t in edges))
# the host's launch order: segments by compressed length in classes of 2^shift bytes, longest first, a class in stream order
shift = int(os.environ.get("QATZIP_AMD_INFLATE_CLS", "5"))
cls = np.minimum(lens.astype(np.int64) >> shift, 8191)
order = np.argsort(-cls, kind="stable")
ol = lens[order].astype(np.float64)
wl = np.array([ol[w * spw:(w + 1) * spw].max() for w in range(nw)], np.float64)
ws = np.array([ol[w * spw:(w + 1) * spw].sum() for w in range(nw)], np.float64)
span = np.array([np.ptp(order[w * spw:(w + 1) * spw]) for w in range(nw)], np.float64)
print("  correlation of a wave's duration with its longest segment's input %.3f, with its segments' input together %.3f, with its number %.3f" %
      (np.corrcoef(dur, wl)[0, 1], np.corrcoef(dur, ws)[0, 1], np.corrcoef(dur, np.arange(nw))[0, 1]))
print("  a wave's segments lie p50 %.0f / max %.0f segments apart in the stream; ms per compressed KB of a wave's longest segment: p10 %.3f p50 %.3f p90 %.3f" %
      (float(np.percentile(span, 50)), span.max(), *[float(np.percentile(dur / (wl / 1024), p)) for p in (10, 50, 90)]))
late = np.argsort(e)[-16:]
print("  the sixteen waves that end last: " + "  ".join("w%d %.1f-%.1f" % (w, b[w], e[w]) for w in late))
first = b < 1.0
print("  waves started in the first ms: %d, their duration p50 %.2f max %.2f; the others: p50 %.2f max %.2f" %
      (int(first.sum()), float(np.percentile(dur[first], 50)), dur[first].max(), float(np.percentile(dur[~first], 50)), dur[~first].max()))
