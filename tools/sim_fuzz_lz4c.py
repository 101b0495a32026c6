#!/usr/bin/env python3
"""Developer tool: randomized campaign on the CPU emulator for the LZ4 block compressor of the pull kernel
(qzk_lz4_block_f: content-carrying table entries under an epoch, register window, LDS-staged output): frames of random kinds /
sizes / frame sizes by one to three persistent waves against the oracle, byte for byte.  QZSIM_LZ4_EPOCH0=65500 starts the
waves' epochs at the 16-bit wrap.  usage: sim_fuzz_lz4c.py [seconds] [first seed]"""
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402

S = C.CDLL(os.environ.get("QZSIM_SO") or os.path.join(ROOT, "tests", "sim", "libqzsim.so"))
S.sim_lz4c_pull.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n_ok = 0; nfr_tot = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(datagen.KINDS)
    fs = rng.choice([65536, 65536, 16384, 4096, 1024, 100, 65535, 13])
    n = rng.choice([rng.randrange(0, 300), rng.randrange(300, 20000), rng.randrange(20000, 140000)])
    if fs <= 100:
        n = min(n, 3000)
    if kind == "lzmix":
        n = min(n, 30000)
    src = datagen.gen_bytes(kind, n, 3000 + seed)
    if rng.random() < 0.3 and n > 64:
        cut = rng.randrange(1, n); src = (src[cut:] + src[:cut] + src)[:n]
    if rng.random() < 0.2 and n > 1000:
        at = rng.randrange(0, n - 500); per = bytes(rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 5, 7, 8, 13, 300])))
        ln = rng.randrange(10, min(n - at, 40000)); src = src[:at] + (per * (ln // len(per) + 1))[:ln] + src[at + ln:]
    nfr = max(1, (n + fs - 1) // fs)
    stride = (fs + 15 + 4 + 8 + 64 + 15) & ~15
    slots = np.zeros(nfr * stride + 64, np.uint8); lens = np.zeros(nfr, np.uint32)
    off = rng.randrange(0, 16)                                    # the slots at any 16-byte phase
    S.sim_lz4c_pull(src, n, fs, slots.ctypes.data + off, stride, lens.ctypes.data, rng.choice([1, 2, 3]))
    ok = True
    for i in range(nfr):
        piece = src[i * fs:(i + 1) * fs]
        exp = O.sw_compress("LZ4", piece, 65536, 1, cap=len(piece) + len(piece) // 255 + 200)[2]
        if bytes(slots[off + i * stride:off + i * stride + int(lens[i])]) != exp:
            ok = False; bad.append((seed, kind, n, fs, i)); print("MISMATCH", bad[-1], flush=True); break
    n_ok += ok; nfr_tot += nfr
    seed += 1
print("up to seed %d: %d calls ok (%d frames), %d mismatches %s" % (seed - 1, n_ok, nfr_tot, len(bad), bad))
