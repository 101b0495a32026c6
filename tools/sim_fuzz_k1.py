#!/usr/bin/env python3
"""Developer tool: level-1 (K1 + K2) campaign on the CPU emulator with the sizes that stress the candidate table:
chunks up to 512 KB (several window slides per chunk, chunk-absolute offsets against the window origin), many chunks per
run (the table column of a wave is reused dirty, chunk after chunk, told apart by epochs only), matches at distances
around MAX_DIST.  usage: sim_fuzz_k1.py [seconds] [first seed] [wide]   (wide: the experimental K1w, chunks of at most 64 KB)"""
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
S.sim_deflate_wide.argtypes = S.sim_deflate_fused.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
t0 = time.time(); n_ok = 0; bad = []
seed = seed0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(["silesia", "text", "records", "runs", "mod200", "allA", "rand", "lzmix"])
    chunk = rng.choice([16384, 65536, 65536, 131072, 262144, 524288])
    if WIDE:
        chunk = rng.choice([4096, 16384, 65536, 65536, 65536])
    n = rng.choice([rng.randrange(60000, 140000), rng.randrange(120000, 400000), rng.randrange(300000, 700000)])
    if kind == "lzmix":
        n = min(n, 90000)
    src = bytearray(datagen.gen_bytes(kind, n, 5000 + seed))
    for _ in range(rng.randrange(0, 12)):                        # copies at the distances where zlib's rules bite
        d = rng.choice([32506, 32505, 32507, 32768, 32767, 65536, 65274, 1, 258, rng.randrange(1, 40000)])
        ln = rng.choice([3, 4, 5, 8, 9, 16, 17, 258, 300])
        at = rng.choice([32768, 65274, 65536, 98042, 131072]) + rng.randrange(-300, 300) if rng.random() < 0.6 else rng.randrange(0, n)
        if at - d >= 0 and at + ln <= n:
            src[at:at + ln] = src[at - d:at - d + ln]
    src = bytes(src)
    last = rng.choice([1, 1, 0])
    nch = max(1, (n + chunk - 1) // chunk)
    cap = n * 9 // 8 + 4096 * (nch + 1)
    out = C.create_string_buffer(cap); ol = C.c_uint64(0); crcs = np.zeros(nch, np.uint32)
    (S.sim_deflate_wide if WIDE else S.sim_deflate_fused)(src, n, chunk, last, out, C.byref(ol), crcs.ctypes.data)      # the product's shape: K2 and the CRC inside the K1 waves
    exp = O.sw_compress("RAW", src, chunk, 1, last=last, cap=cap)[2]
    import zlib
    crc_ok = all(int(crcs[i]) == (zlib.crc32(src[i * chunk:(i + 1) * chunk]) & 0xffffffff) for i in range(nch))
    if out.raw[:ol.value] != exp or not crc_ok:
        bad.append((seed, kind, n, chunk, last)); print("MISMATCH", bad[-1], flush=True)
    else:
        n_ok += 1
    seed += 1
print("seeds %d..%d: %d ok, %d mismatches %s" % (seed0, seed - 1, n_ok, len(bad), bad))
