#!/usr/bin/env python3
"""Developer tool: device memory one process holds after each kind of call (hipMemGetInfo through ctypes; run it alone on the
device) - what the reference's fleet shape multiplies by the number of processes.  usage: mem_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
from qatzip_amd import api as A  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def used():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return (t.value - f.value) / 2**20


base = used()
print("%-58s %9.1f MiB in use on the device" % ("HIP runtime up", base))
data = datagen.gen_bytes("silesia", 64 << 20, 7)
s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
print("%-58s %9.1f MiB" % ("session set up (context, streams)", used()))
last = None
for what, n in (("qzCompress of 64 KB", 65536), ("qzCompress of 512 KB (the harness' default buffer)", 524288),
                ("qzCompress of 16 MiB", 16 << 20), ("qzCompress of 64 MiB", 64 << 20)):
    rc, used_in, comp, _ = s.compress(data[:n])
    assert rc == 0 and used_in == n
    print("%-58s %9.1f MiB" % (what, used()))
    rc, _, out = s.decompress(comp, n + 64)[:3]
    assert rc == 0 and out == data[:n]
    print("%-58s %9.1f MiB" % (what.replace("qzCompress", "qzDecompress"), used()))
rc, _, comp, _ = s.compress(data[:524288]); s.decompress(comp, 524288 + 64)
print("%-58s %9.1f MiB" % ("a 512 KB pair again (pools shrink?)", used()))
s.close()
print("%-58s %9.1f MiB" % ("session torn down", used()))
