#!/usr/bin/env python3
"""Developer tool: -DQZK_PROF build of the library, the workgroup-per-chunk parse (QATZIP_AMD_K1=wide, qzk_deflate_wide.h)
over a few hundred chunks, cycles per phase and window (QATZIP_AMD_WIDE_PROF=1: the kernel's run-time clocks).
usage: prof_wide.py [kind] [chunks]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402

SO = os.environ.get("QATZIP_AMD_SO") or os.path.join(ROOT, "qatzip_amd", "libqatzip_amd.so")     # the product library: the clocks are a run-time option


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "silesia"
    nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    os.environ["QATZIP_AMD_K1"] = "wide"
    os.environ["QATZIP_AMD_WIDE_PROF"] = "1"
    L = C.CDLL(SO)
    vp = C.c_void_p
    L.qzd_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.qzd_dev_alloc.argtypes = [vp, C.c_size_t]; L.qzd_dev_alloc.restype = vp
    L.qzd_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.qzd_deflate_raw.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64), vp]
    L.qzd_debug_wide_prof.argtypes = [vp, vp, C.c_uint32]
    L.qzd_last_timing.argtypes = [vp, C.POINTER(C.c_float * 4)]
    h = vp(); assert L.qzd_create(0, C.byref(h)) == 0
    n = nchunks * 65536
    src = datagen.gen(kind, n, 7)
    d_src = L.qzd_dev_alloc(h, n + 512); d_dst = L.qzd_dev_alloc(h, n * 2)
    L.qzd_h2d(h, d_src, src.ctypes.data, n)
    ol = C.c_uint64(0)
    for _ in range(2):
        assert L.qzd_deflate_raw(h, d_src, n, 65536, 1, 1, d_dst, n * 2, C.byref(ol), None) == 0
    ms = (C.c_float * 4)(); L.qzd_last_timing(h, C.byref(ms))
    buf = np.zeros(nchunks * 16, np.uint64)
    got = L.qzd_debug_wide_prof(h, buf.ctypes.data, nchunks)
    assert got == nchunks, got
    prof = buf.reshape(nchunks, 16).astype(np.float64)
    tot = prof.mean(0)
    names = {12: "chunk load / head clear", 0: "sort pass 1 (+hash, gather issue)", 1: "sort pass 2", 2: "group structure, chain, 16-byte compares",
             3: "selection (ballots, carry)", 4: "check (+ window tail)", 5: "matches, round 0 -> LDS", 13: "matches, later rounds", 6: "extension", 7: "doubling in the wave",
             8: "exit chain, inserted set", 9: "final prefix: symbols, chains"}
    win, rounds = tot[10], tot[11]
    print("kind=%s chunks=%d ratio=%.3f  parse kernel %.2f ms  K2+CRC %.2f ms  -> %.2f GB/s parse-only"
          % (kind, nchunks, ol.value / n, ms[0], ms[1], n / (ms[0] * 1e-3) / 1e9))
    print("  windows/chunk %.1f  rounds/window %.2f  cycles/chunk %.0f (%.1f us at 2.4 GHz)"
          % (win, rounds / win, tot[14], tot[14] / 2400))
    for k in (12, 0, 1, 2, 3, 4, 5, 13, 6, 7, 8, 9):
        print("  %-42s %10.0f cycles/chunk %5.1f %%  %8.0f /window" % (names[k], tot[k], 100 * tot[k] / tot[14], tot[k] / win))


if __name__ == "__main__":
    main()
