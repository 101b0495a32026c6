"""Small requests, two ways: a loop of synchronous qzCompress calls, and the same requests through qzCompress2 with
everything in flight (the submission queue coalesces what is waiting into one launch).  python tools/async_bench.py [n] [bytes]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
from qatzip_amd import api as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
big = datagen.gen_bytes("silesia", 8 << 20, 7)
srcs = [big[(i * 37717) % (len(big) - sz):][:sz] for i in range(n)]
s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
L = s.L
s.compress(srcs[0], 1)
m = min(n, 200)
t = time.time()
for x in srcs[:m]:
    rc, used, out, _ = s.compress(x, 1)
    assert rc == 0
dt = time.time() - t
print("sync  : %5d x %d B  %.3f s  %.1f req/s  %.3f GB/s" % (m, sz, dt, m / dt, m * sz / dt / 1e9))

ins = [C.create_string_buffer(x, len(x)) for x in srcs]
outs = [C.create_string_buffer(sz * 9 // 8 + 1024) for _ in srcs]
res = [A.QzResult() for _ in srcs]
done = threading.Event(); cnt = [0]


def cbf(r):
    cnt[0] += 1
    if cnt[0] == n:
        done.set()
    return 0


cb = A.QzAsyncCallback(cbf)
t = time.time()
for i in range(n):
    res[i].src_len = sz; res[i].dest_len = len(outs[i])
    assert L.qzCompress2(C.byref(s.s), ins[i], outs[i], cb, C.byref(res[i])) == 0
assert done.wait(600)
dt = time.time() - t
assert all(r.status == 0 for r in res)
a, b = C.c_uint64(), C.c_uint64()
L.qzamd_async_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
L.qzamd_async_stats(C.byref(a), C.byref(b))
print("async : %5d x %d B  %.3f s  %.1f req/s  %.3f GB/s  (%d launches carried %d requests)" % (n, sz, dt, n / dt, n * sz / dt / 1e9, a.value, b.value))
s.close()
