#!/usr/bin/env python3
"""Developer tool: randomized decode campaign on the CPU emulator - deflate streams from zlib at random level / strategy /
flush pattern, decoded by K3 (one wave straight through, and per flush segment) and by the two-phase lane decoder.
usage: sim_fuzz_inflate.py [seconds] [first seed]"""
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
so = os.environ.get("QZSIM_SO") or os.path.join(SIMDIR, "libqzsim.so")       # QZSIM_SO: an emulator build with other -D flags
if not os.path.exists(so):
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-I", SIMDIR, "-Wno-unused-function", "-o", so,
                           os.path.join(SIMDIR, "sim_driver.cpp")] + os.environ.get("QZSIM_FLAGS", "").split())
S = C.CDLL(so)
for f in (S.sim_inflate, S.sim_inflate_lane):
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
S.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
handed_back = [0, 0]


def spec(K):
    def fn(comp, out, segs, res, n):
        handed_back[0] += S.sim_inflate_spec(comp, out, segs, res, n, K); handed_back[1] += n
    return fn
seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])


def run(fn, comp, segs, n):
    cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.full(n + 64, 0xAA, np.uint8)
    sa = np.array(segs, dtype=seg_dt); res = np.zeros(len(segs), res_dt)
    fn(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs))
    return bytes(obuf[:n]), res, bytes(obuf[n:n + 64])


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n_ok = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(datagen.KINDS)
    n = rng.choice([rng.randrange(1, 300), rng.randrange(300, 20000), rng.randrange(20000, 120000)])
    if kind == "lzmix":
        n = min(n, 30000)
    src = datagen.gen_bytes(kind, n, 5000 + seed)
    if rng.random() < 0.3 and n > 64:
        cut = rng.randrange(1, n); src = (src[cut:] + src[:cut] + src)[:n]
    level = rng.choice([0, 1, 3, 6, 9])
    strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
    co = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1, 8, 9]), strat)
    pieces, cuts, pos = [], [], 0
    full = rng.random() < 0.5                               # full flushes: independent segments; sync flushes: history carries over
    while pos < n:
        k = min(n - pos, rng.choice([n, 70000, 16384, 4096, 1000]))
        pieces.append(co.compress(src[pos:pos + k]))
        pos += k
        if pos < n:
            pieces[-1] += co.flush(zlib.Z_FULL_FLUSH if full else zlib.Z_SYNC_FLUSH)
        cuts.append(k)
    pieces[-1] += co.flush()
    comp = b"".join(pieces)
    ok = True
    got, res, tail = run(S.sim_inflate, comp, [(0, 0, len(comp), n, 2, 0)], n)            # straight through
    ok &= got == src and res[0]["status"] == 0 and res[0]["in_used"] == len(comp) and tail == b"\xaa" * 64
    if full:                                                # per segment, both decoders
        segs, io, oo = [], 0, 0
        for pc, k in zip(pieces, cuts):
            segs.append((io, oo, len(comp) - io, k, 0, len(pc))); io += len(pc); oo += k
        for fn in (S.sim_inflate, S.sim_inflate_lane, spec(rng.choice([2, 4, 8, 16, 32]))):
            got, res, tail = run(fn, comp, segs, n)
            ok &= got == src and bool((res["status"] >= 0).all()) and tail == b"\xaa" * 64
            ok &= [int(r["in_used"]) for r in res] == [len(pc) for pc in pieces]
    if not ok:
        bad.append((seed, kind, n, level, strat, full)); print("MISMATCH", bad[-1], flush=True)
    else:
        n_ok += 1
    seed += 1
print("up to seed %d: %d ok, %d mismatches %s; K lanes per segment: %d of %d segments handed back to the serial kernel" %
      (seed - 1, n_ok, len(bad), bad, handed_back[0], handed_back[1]))
