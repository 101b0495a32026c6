#!/usr/bin/env python3
"""qzCompress from pinned host memory, 1 GiB per call, under different splits of the host-input pipeline
(QATZIP_AMD_HOST_BATCH / QATZIP_AMD_HOST_FIRST: chunks per batch / in the first batch).  Prints GB/s per setting."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import datagen                      # noqa: E402
from qatzip_amd import api as A     # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mb << 20
L = A.lib()
s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
cap = L.qzMaxCompressedLength(n, C.byref(s.s)) + 64
p_src, p_dst = L.qzMalloc(n, 0, A.PINNED_MEM), L.qzMalloc(cap, 0, A.PINNED_MEM)
base = datagen.gen("silesia", 64 << 20, 1)
hsrc = np.ctypeslib.as_array((C.c_ubyte * n).from_address(p_src))
for off in range(0, n, len(base)):
    k = min(len(base), n - off)
    hsrc[off:off + k] = base[:k]
ref = None
for batch, first in [(0, 0), (8192, 8192), (8192, 4096), (6144, 4096), (4096, 4096), (4096, 2048), (12288, 4096), (16384, 16384), (5462, 5462)]:
    for k, v in (("QATZIP_AMD_HOST_BATCH", batch), ("QATZIP_AMD_HOST_FIRST", first)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    best = 1e9
    for it in range(3):
        sl, dl = C.c_uint(n), C.c_uint(cap)
        t0 = time.perf_counter()
        rc = L.qzCompress(C.byref(s.s), C.cast(p_src, C.c_char_p), C.byref(sl), p_dst, C.byref(dl), 1)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0 and sl.value == n, rc
    out = bytes((C.c_ubyte * dl.value).from_address(p_dst))
    import zlib
    h = zlib.crc32(out)
    ref = ref if ref is not None else h
    print("batch %5d first %5d: %.2f GB/s (%.1f ms) out %d %s" % (batch, first, n / best / 1e9, best * 1e3, dl.value, "same" if h == ref else "DIFFERENT"), flush=True)
