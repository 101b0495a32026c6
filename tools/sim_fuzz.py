#!/usr/bin/env python3
"""Developer tool: randomized parity campaign on the CPU emulator - the kernel headers against the oracle on inputs the
fixed tests do not use (random kind / size / chunk size / level / seed).  usage: sim_fuzz.py [seconds] [first seed]"""
import ctypes as C
import os
import random
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402

SIMDIR = os.path.join(ROOT, "tests", "sim")
so = os.path.join(SIMDIR, "libqzsim.so")
if not os.path.exists(so):
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-I", SIMDIR, "-Wno-unused-function", "-o", so,
                           os.path.join(SIMDIR, "sim_driver.cpp")])
S = C.CDLL(so)
A8 = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
S.sim_deflate.argtypes = A8
S.sim_deflate_level.argtypes = A8[:4] + [C.c_int] + A8[4:]
S.sim_deflate_lazy.argtypes = S.sim_deflate_level.argtypes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n_ok = 0; bad = []
seed = seed0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(datagen.KINDS)
    chunk = rng.choice([1024, 2048, 4096, 16384, 65536, 131072])
    n = rng.choice([rng.randrange(0, 300), rng.randrange(300, 20000), rng.randrange(20000, 90000), rng.randrange(60000, 140000)])
    if kind == "lzmix":
        n = min(n, 30000)
    level = rng.choice([1, 1, 1, 2, 3, 4, 5, 6, 9])
    if level >= 4:
        n = min(n, 40000 if chunk <= 65536 else 70000)          # the emulator meets at every hop
    last = rng.choice([1, 1, 0])
    src = datagen.gen_bytes(kind, n, 1000 + seed)
    if rng.random() < 0.3 and n > 64:                           # splice: repeats at long distances, mixed statistics
        cut = rng.randrange(1, n)
        src = (src[cut:] + src[:cut] + src)[:n]
    nch = max(1, (n + chunk - 1) // chunk)
    cap = n * 9 // 8 + 4096 * (nch + 1)
    out = C.create_string_buffer(cap); ol = C.c_uint64(0); crcs = np.zeros(nch, np.uint32)
    if level == 1:
        S.sim_deflate(src, n, chunk, last, out, C.byref(ol), crcs.ctypes.data)
    elif level < 4:
        S.sim_deflate_level(src, n, chunk, last, level, out, C.byref(ol), crcs.ctypes.data)
    else:
        S.sim_deflate_lazy(src, n, chunk, last, level, out, C.byref(ol), crcs.ctypes.data)
    exp = O.sw_compress("RAW", src, chunk, level, last=last, cap=cap)[2]
    if out.raw[:ol.value] != exp:
        bad.append((seed, kind, n, chunk, level, last)); print("MISMATCH", bad[-1], flush=True)
    else:
        n_ok += 1
    seed += 1
print("seeds %d..%d: %d ok, %d mismatches %s" % (seed0, seed - 1, n_ok, len(bad), bad))
