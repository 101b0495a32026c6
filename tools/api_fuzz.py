#!/usr/bin/env python3
"""Developer tool: randomized campaign through qatzip.h on the GPU - random format / hw_buff_sz / level / sizes, streams
opened with last = 0 and closed later, several members per buffer, and a caller's decompress loop with destinations from
far too small to ample (whole members, pieces of members, flow control by QZ_BUF_ERROR).  usage: api_fuzz.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402
from qatzip_amd import api as A  # noqa: E402

FMT = {"4B": A.QZ_DEFLATE_4B, "GZIP": A.QZ_DEFLATE_GZIP, "GZIP_EXT": A.QZ_DEFLATE_GZIP_EXT, "RAW": A.QZ_DEFLATE_RAW}
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time(); n_ok = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    fmt = rng.choice(["GZIP_EXT", "GZIP_EXT", "GZIP", "RAW", "4B", "ZLIB", "LZ4"])
    hw = rng.choice([16384, 65536, 65536, 131072])
    lvl = rng.choice([1, 1, 1, 4, 2])
    if fmt == "LZ4":
        lvl = 1
    s = (A.Session(hw_buff_sz=hw, comp_lvl=lvl, zlib_format=True) if fmt == "ZLIB" else
         A.Session(hw_buff_sz=hw, lz4=True) if fmt == "LZ4" else A.Session(FMT[fmt], hw, comp_lvl=lvl))
    ok = s.rc_setup == A.QZ_OK
    hwf = fmt in ("GZIP_EXT", "GZIP") and lvl == 1 and rng.random() < 0.25
    if hwf:
        import ctypes as C
        s.L.qzamd_set_hw_framing.argtypes = [C.c_void_p, C.c_int]
        ok &= s.L.qzamd_set_hw_framing(C.byref(s.s), 1) == A.QZ_OK
    members, plain = [], []
    for m in range(rng.choice([1, 1, 2, 4]) if fmt not in ("RAW", "4B") else 1):
        kind = rng.choice(datagen.KINDS)
        n = rng.choice([rng.randrange(0, 200), rng.randrange(200, 70000), rng.randrange(70000, 900000)])
        if kind == "lzmix":
            n = min(n, 100000)
        src = datagen.gen_bytes(kind, n, 20000 + seed * 7 + m)
        if fmt == "LZ4":                                        # one frame per call: one block, or liblz4's linked blocks above 64 KB
            rc, used, out, _ = s.compress(src, 1, cap=n + 64 * (n // 65536 + 2))
            exp = O.sw_compress("LZ4", src, 65536, 1, cap=n + 64 * (n // 65536 + 2))[2]
            ok &= rc == A.QZ_OK and used == n and out == exp
        elif hwf and n >= 1024:                                 # the hardware path's framing: a complete member per chunk (calls below
                                                                # input_sz_thrshold keep the software path's, src/qatzip.c:1934-1947)
            import zlib
            rc, used, out, _ = s.compress(src, 1)
            exp = b""
            for off in range(0, n, hw):
                ch = src[off:off + hw]
                body = O.sw_compress("RAW", ch, hw, lvl, last=1)[2]
                hdr = (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0]) + b"QZ" + (8).to_bytes(2, "little") +
                       len(ch).to_bytes(4, "little") + len(body).to_bytes(4, "little")) if fmt == "GZIP_EXT" else bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255])
                exp += hdr + body + (zlib.crc32(ch) & 0xffffffff).to_bytes(4, "little") + len(ch).to_bytes(4, "little")
            ok &= rc == A.QZ_OK and used == n and out == exp
        elif rng.random() < 0.3 and n > hw and fmt != "4B":     # the member written by two calls: last = 0, then last = 1
            # (not for the 4-byte header: a stream opened with last = 0 keeps a zero length there, src/qatzip_sw.c:166)
            cut = (rng.randrange(1, n) // hw) * hw or hw
            rc1, u1, o1, _ = s.compress(src[:cut], 0)
            rc2, u2, o2, _ = s.compress(src[cut:], 1)
            ok &= rc1 == A.QZ_OK and rc2 == A.QZ_OK and u1 == cut and u2 == n - cut
            out = o1 + o2
            exp = O.sw_compress(fmt, src[:cut], hw, lvl, last=0, cap=n * 9 // 8 + 65536)[2]      # the opening call alone is checkable
            ok &= o1 == exp
        else:
            rc, used, out, _ = s.compress(src, 1)
            ok &= rc == A.QZ_OK and used == n and out == O.sw_compress(fmt, src, hw, lvl, cap=n * 9 // 8 + 65536)[2]
        members.append(out); plain.append(src)
    comp, want = b"".join(members), b"".join(plain)
    # the caller's loop with a destination of random size
    cap = max(rng.choice([64, 1000, 20000, hw, hw + 1, 300000, len(want) + 16]), len(want) // 1500 + 1)    # a few thousand calls at most
    got, pos, calls = b"", 0, 0
    while pos < len(comp) and calls < 20000:
        rc, used, back = s.decompress(comp[pos:], cap)
        calls += 1
        if rc == A.QZ_BUF_ERROR and not used and not back:
            cap *= 2                                            # nothing fits: grow, like utils/qzip.c
            continue
        if rc not in (A.QZ_OK, A.QZ_BUF_ERROR) or not (used or back):
            ok = False; break
        got += back; pos += used
    ok &= got == want and pos == len(comp)
    s.close()
    if not ok:
        bad.append((seed, fmt, hw, lvl, [len(p) for p in plain], cap)); print("MISMATCH", bad[-1], flush=True)
    else:
        n_ok += 1
    seed += 1
print("up to seed %d: %d ok, %d mismatches %s" % (seed - 1, n_ok, len(bad), bad))
