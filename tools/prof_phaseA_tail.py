#!/usr/bin/env python3
"""Developer tool: phase A's per-SEGMENT finishing times against its per-wave times (a -DQZK_SPEC_PROF build:
QATZIP_AMD_SO=build/var/lib_sprof.so) - the histogram VERDICT r4 item 2 asked for.  A segment's time = shader clocks from its
wave's start to the moment its lane 0 had its last block (rounds are wave-wide: a segment also waits for its wave-mates'
rounds).  usage: prof_phaseA_tail.py MiB[:chunkKiB] ..."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402

shapes = [tuple(int(x) for x in (a.split(":") + ["64"])[:2]) for a in (sys.argv[1:] or ["64", "256", "4096"])]
top = max(mb for mb, _ in shapes) << 20
base = datagen.gen("silesia", min(128 << 20, top), 20250523)
ctx = qatzip_amd.Context(0)
d_src = ctx.alloc(top)
P = len(base) - 4099 if top > len(base) else len(base)
for off in range(0, top, P):
    d_src.upload(base[:min(P, top - off)], off)
d_c = ctx.alloc(qatzip_amd.max_deflate_len(top, 16384)); d_o = ctx.alloc(top)
ctx.L.qzd_spec_seg_prof.argtypes = [C.c_void_p, C.c_uint32]
for mb, ck in shapes:
    n = mb << 20
    ctx.deflate_raw_async(d_src, n, ck << 10, 1, 1, d_c); ctx.sync()
    clen = ctx.result()
    ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=False)
    assert ctx.L.qzd_spec_prof(None, C.c_uint32(0)) == 0
    ctx.inflate_stream(d_c, clen, d_o, ck << 10, want_crc=False)
    ms = ctx.inflate_timing()
    nseg = min(n // (ck << 10), 1 << 17)
    K = 16 if nseg <= 8192 else 8 if nseg <= 16384 else 4
    spw = 64 // K
    seg = np.zeros(nseg, np.uint32)
    assert ctx.L.qzd_spec_seg_prof(seg.ctypes.data, nseg) == 0
    seg = seg.astype(np.float64) * 64 / 1e6                     # M clocks
    nw = min(8192, (nseg + spw - 1) // spw)
    buf = np.zeros((nw, 8), np.uint64)
    assert ctx.L.qzd_spec_prof(buf.ctypes.data_as(C.c_void_p), C.c_uint32(nw)) == 0
    wav = buf[:, :3].astype(np.float64).sum(1) / 1e6                # (column 3 holds the looks at trails: tools/prof_spec.py)
    q = lambda a, p: float(np.percentile(a, p))
    print("%d MiB / %d KiB segments: %d segments, K = %d lanes each, %d waves; phase A %.2f ms (HIP events)" % (mb, ck, nseg, K, nw, ms[0] - ms[2]))
    print("  per segment (M clocks from its wave's start to its last block): p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (q(seg, 50), q(seg, 90), q(seg, 99), seg.max()))
    print("  per wave (M clocks, the first %d waves):                         p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (nw, q(wav, 50), q(wav, 90), q(wav, 99), wav.max()))
    h, e = np.histogram(seg, bins=12)
    print("  histogram of the segments' times: " + "  ".join("%.1f-%.1f M: %d" % (e[i], e[i + 1], h[i]) for i in range(len(h))))
    print("  a wave lasts as long as its slowest segment: wave p50 / segment p50 = %.2f, launch = max over waves = %.2f x the median segment" % (q(wav, 50) / max(q(seg, 50), 1e-9), wav.max() / max(q(seg, 50), 1e-9)), flush=True)
