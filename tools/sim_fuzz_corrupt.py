#!/usr/bin/env python3
"""Developer tool (emulator, no GPU): damaged deflate streams through the decoders - valid segments with a few bits flipped
(most in the first bytes, where the block header lives) or cut short.  A decoder must never crash, hang or write outside
its output; it reports an error, or - when zlib decodes the damaged stream too - the very bytes zlib produces.
usage: sim_fuzz_corrupt.py [seconds] [first seed]"""
import ctypes as C
import os
import random
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402

SIMDIR = os.path.join(ROOT, "tests", "sim")
so = os.environ.get("QZSIM_SO") or os.path.join(SIMDIR, "libqzsim.so")
if not os.path.exists(so):
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-I", SIMDIR, "-Wno-unused-function", "-o", so,
                           os.path.join(SIMDIR, "sim_driver.cpp")])
S = C.CDLL(so)
for f in (S.sim_inflate, S.sim_inflate_lane):
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
S.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])


def run(fn, comp, n):
    cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.full(n + 64, 0xAA, np.uint8)
    sa = np.array([(0, 0, len(comp), n, 0, len(comp))], dtype=seg_dt); res = np.zeros(1, res_dt)
    fn(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, 1)
    return bytes(obuf[:n]), res[0], bytes(obuf[n:n + 64])


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n_err = n_same = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(datagen.KINDS)
    n = rng.choice([rng.randrange(1, 300), rng.randrange(300, 20000), rng.randrange(20000, 70000)])
    if kind == "lzmix":
        n = min(n, 30000)
    src = datagen.gen_bytes(kind, n, 9000 + seed)
    co = zlib.compressobj(rng.choice([1, 1, 6, 9]), zlib.DEFLATED, -15, 9, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY]))
    comp = bytearray(co.compress(src) + co.flush())
    how = rng.random()
    if how < 0.15 and len(comp) > 4:
        comp = comp[:rng.randrange(1, len(comp))]                  # cut short
    else:
        for _ in range(rng.choice([1, 1, 2, 5])):
            at = rng.randrange(0, min(len(comp), 120)) if rng.random() < 0.7 else rng.randrange(0, len(comp))
            comp[at] ^= 1 << rng.randrange(8)
    comp = bytes(comp)
    want = None
    try:
        d = zlib.decompressobj(-15)
        out = d.decompress(comp, n + 1)
        if d.eof and len(out) == n:
            want = out                                              # zlib takes it and it fills the segment exactly
    except zlib.error:
        pass
    for name, fn in (("wave", S.sim_inflate), ("lane", S.sim_inflate_lane),
                     ("k4", lambda a, b, c, d, e: S.sim_inflate_spec(a, b, c, d, e, 4)),
                     ("k16", lambda a, b, c, d, e: S.sim_inflate_spec(a, b, c, d, e, 16))):
        got, r, tail = run(fn, comp, n)
        ok = tail == b"\xaa" * 64
        if r["status"] >= 0 and r["out_len"] == n:
            ok &= want is not None and got == want                  # claimed success: zlib must agree, byte for byte
            n_same += 1
        else:
            n_err += 1
        if not ok:
            bad.append((seed, name, kind, n, int(r["status"]), int(r["out_len"]), want is not None)); print("MISMATCH", bad[-1], flush=True)
    seed += 1
print("up to seed %d: %d decodes ended in an error, %d in zlib's own bytes, %d mismatches %s" % (seed - 1, n_err, n_same, len(bad), bad))
