#!/usr/bin/env python3
"""qzCompress from pinned host memory, launch fed while it runs (qzk_wait_input) against the batched pipeline
(QATZIP_AMD_HOST_BATCHED=1): same bytes?  Every round overwrites the SAME source buffer with different data, so a
stale cache line anywhere would show as a difference.  Prints the rates of both."""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import datagen                      # noqa: E402
from qatzip_amd import api as A     # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = mb << 20
L = A.lib()
s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
cap = L.qzMaxCompressedLength(n, C.byref(s.s)) + 64
p_src, p_dst = L.qzMalloc(n, 0, A.PINNED_MEM), L.qzMalloc(cap, 0, A.PINNED_MEM)
hsrc = np.ctypeslib.as_array((C.c_ubyte * n).from_address(p_src))
kinds = ["silesia", "text", "records", "lzmix", "rand", "runs"]
bad = 0
for r in range(rounds):
    base = datagen.gen(kinds[r % len(kinds)], 32 << 20, 100 + r)
    for off in range(0, n, len(base)):
        k = min(len(base), n - off)
        hsrc[off:off + k] = base[:k]
    res = {}
    for mode in ("stream", "batched", "stream"):
        if mode == "batched":
            os.environ["QATZIP_AMD_HOST_BATCHED"] = "1"
        else:
            os.environ.pop("QATZIP_AMD_HOST_BATCHED", None)
        sl, dl = C.c_uint(n), C.c_uint(cap)
        t0 = time.perf_counter()
        rc = L.qzCompress(C.byref(s.s), C.cast(p_src, C.c_char_p), C.byref(sl), p_dst, C.byref(dl), 1)
        dt = time.perf_counter() - t0
        assert rc == 0 and sl.value == n, rc
        h = zlib.crc32(bytes((C.c_ubyte * dl.value).from_address(p_dst)))
        res.setdefault(mode, []).append((h, dl.value, n / dt / 1e9))
    same = len({x[0] for v in res.values() for x in v}) == 1
    bad += not same
    print("%-8s stream %.2f / %.2f GB/s  batched %.2f GB/s  %d bytes  %s" % (kinds[r % len(kinds)], res["stream"][0][2], res["stream"][1][2],
          res["batched"][0][2], res["stream"][0][1], "same" if same else "DIFFERENT"), flush=True)
print("differences:", bad)
sys.exit(1 if bad else 0)
