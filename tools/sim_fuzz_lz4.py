#!/usr/bin/env python3
"""Developer tool: randomized campaign on the CPU emulator for the LZ4 frame decoder (qzk_lz4d_kernel): frames the oracle
writes for random kinds / sizes (one block, linked blocks, stored blocks), several per launch at random output phases,
a share of them damaged - the source's bytes or an error, never a byte outside the output.
usage: sim_fuzz_lz4.py [seconds] [first seed]"""
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
S.sim_lz4d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
SEG = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4")])
RES = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("pad", "<u4")])
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n_ok = n_err = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    frames, srcs, caps, dmg = [], [], [], []
    for _ in range(rng.choice([1, 1, 2, 5])):
        kind = rng.choice(datagen.KINDS)
        n = rng.choice([rng.randrange(0, 300), rng.randrange(300, 20000), rng.randrange(20000, 70000), rng.randrange(65537, 200000)])
        if kind == "lzmix":
            n = min(n, 30000)
        src = datagen.gen_bytes(kind, n, 7000 + seed)
        if rng.random() < 0.3 and n > 64:
            cut = rng.randrange(1, n); src = (src[cut:] + src[:cut] + src)[:n]
        if rng.random() < 0.15 and n > 1000:                       # runs and short periods spliced in
            at = rng.randrange(0, n - 500); per = bytes(rng.randrange(256) for _ in range(rng.choice([1, 2, 3, 5, 7, 8, 13])))
            ln = rng.randrange(10, min(n - at, 40000)); src = src[:at] + (per * (ln // len(per) + 1))[:ln] + src[at + ln:]
        fr = bytearray(O.sw_compress("LZ4", src, 65536, 1, cap=n + n // 255 + 4096)[2])
        d = rng.random() < 0.2
        if d:
            if rng.random() < 0.3 and len(fr) > 8:
                fr = fr[:rng.randrange(7, len(fr))]
            else:
                for _ in range(rng.choice([1, 2, 4])):
                    fr[rng.randrange(0, len(fr))] ^= 1 << rng.randrange(8)
        frames.append(bytes(fr)); srcs.append(src); caps.append(n); dmg.append(d)
    lead = rng.randrange(0, 17)
    comp = b"".join(frames)
    cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy()
    total = lead + sum(caps)
    obuf = np.full(total + 64, 0xAA, np.uint8)
    segs, io, oo = [], 0, lead
    for f, cap in zip(frames, caps):
        segs.append((io, oo, len(f), cap)); io += len(f); oo += cap
    sa = np.array(segs, dtype=SEG); res = np.zeros(len(segs), RES)
    S.sim_lz4d(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs))
    ok = bytes(obuf[:lead]) == b"\xaa" * lead and bytes(obuf[total:]) == b"\xaa" * 64
    oo = lead
    for i, (src, cap, d) in enumerate(zip(srcs, caps, dmg)):
        got = bytes(obuf[oo:oo + cap]); oo += cap
        if res[i]["status"] == 0:
            ok &= got == src and res[i]["out_len"] == cap
        else:
            ok &= d; n_err += 1
    if not ok:
        bad.append(seed); print("MISMATCH seed", seed, flush=True)
    else:
        n_ok += 1
    seed += 1
print("up to seed %d: %d launches ok (%d damaged frames ended in an error), %d mismatches %s" % (seed - 1, n_ok, n_err, len(bad), bad))
