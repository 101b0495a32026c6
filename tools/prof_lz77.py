#!/usr/bin/env python3
"""Developer tool: build a -DQZK_PROF variant of the library and print K1's per-phase cycle breakdown."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402

SO = os.path.join(ROOT, "build", "var", "lib_k1prof.so")       # built here (python tools/prof_lz77.py build), travels with the snapshot


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DQZK_PROF",
                           "-Wno-unused-value", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "qatzip_amd", "csrc", "qzd_device.hip"), "-o", SO])


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "silesia"
    nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    if not os.path.exists(SO):
        build()
    L = C.CDLL(SO)
    vp = C.c_void_p
    L.qzd_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.qzd_dev_alloc.argtypes = [vp, C.c_size_t]; L.qzd_dev_alloc.restype = vp
    L.qzd_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.qzd_deflate_raw.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp, C.c_uint64,
                                  C.POINTER(C.c_uint64), vp]
    L.qzd_debug_meta.argtypes = [vp, C.c_int, vp, C.c_uint32]
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
    buf = np.zeros(nchunks * 64, np.uint64)
    sz = L.qzd_debug_meta(h, 0, buf.ctypes.data, nchunks)
    m = buf.view(np.uint8)[:nchunks * sz].reshape(nchunks, sz)
    raw = m[:, sz - 192:].copy().view(np.uint64).reshape(nchunks, 24)
    hits = (raw[:, 8] >> np.uint64(32)).astype(np.float64)          # round 6: hits of the LDS entry cache ride in the window count's high half
    raw[:, 8] &= np.uint64(0xffffffff)
    asked = (raw[:, 15] >> np.uint64(32)).astype(np.float64)        # windows that had to ask the table themselves (entries asked ahead: none / too few)
    raw[:, 15] &= np.uint64(0xffffffff)
    prof = raw.astype(np.float64)
    tot = prof.mean(0)
    names = ["top/tail", "own loads", "cand. loads issued", "cand compares", "slot detect", "serial hops", "exact path",
             "epilogue syms", "windows", "complex lanes", "suspect commits", "symbols", "commit", "", "", "",
             "entry lookup", "wait (asked ahead)", "deferred stores"]
    PH = (0, 17, 18, 1, 16, 2, 3, 4, 5, 6, 7, 12)
    cyc = sum(tot[k] for k in PH)
    print("kind=%s chunks=%d ratio=%.3f  lz77 %.2f ms huff %.2f ms  (clock ticks below are s_memtime units @100MHz?)"
          % (kind, nchunks, ol.value / n, ms[0], ms[1]))
    for k in PH:
        print("  %-16s %12.0f ticks/chunk  %5.1f %%  %8.1f /window" % (names[k], tot[k], 100 * tot[k] / cyc, tot[k] / tot[8]))
    print("  windows/chunk %.0f  complex lanes/window %.2f  suspect commits/window %.2f  symbols/window %.1f"
          % (tot[8], tot[9] / tot[8], tot[10] / tot[8], tot[11] / tot[8]))
    print("  LDS entry cache: %.1f hits/window (of 64 lookups); windows that asked the table themselves: %.1f %%" % (hits.mean() / tot[8], 100 * asked.mean() / tot[8]))
    print("  exact path: %.2f lanes/window, of which %.2f leave at the first test (no earlier lane with the hash, nothing to extend)"
          % (tot[9] / tot[8], tot[15] / tot[8]))
    print("  K2 in the wave: %.0f ticks/chunk (%.1f %% on top of the parse), of which the serial tree build %.0f"
          % (tot[13], 100 * tot[13] / cyc, tot[14]))
    per = prof[:, list(PH)].sum(1)
    print("  total ticks/chunk %.0f  (min %.0f  max %.0f)  => ideal %.2f ms at 256 CUs, 2.3 GHz"
          % (cyc, per.min(), per.max(), per.sum() / 256 / 2.3e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main()
