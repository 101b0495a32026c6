#!/usr/bin/env python3
"""Developer tool: randomized parity campaign on the GPU through the device C ABI - compress at a random level / chunk size /
size / kind against the oracle, then inflate the result (and zlib streams of random level / strategy / flush pattern) back.
usage: gpu_fuzz.py [seconds] [first seed]"""
import os
import random
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402
import qatzip_amd  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = qatzip_amd.Context(0)
CAP = 3 << 20
d_src = ctx.alloc(CAP + 4096); d_c = ctx.alloc(qatzip_amd.max_deflate_len(CAP, 1024) + 4096); d_o = ctx.alloc(CAP + 4096)
t0 = time.time(); n_ok = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(datagen.KINDS)
    chunk = rng.choice([1024, 4096, 16384, 65536, 65536, 131072, 524288])
    n = rng.choice([rng.randrange(0, 300), rng.randrange(300, 70000), rng.randrange(70000, 400000), rng.randrange(400000, 2_500_000)])
    if kind == "lzmix":
        n = min(n, 140000)
    level = rng.choice([1, 1, 1, 1, 2, 3, 4, 5, 6, 7, 9])
    if level >= 6:
        n = min(n, 600000)
    last = rng.choice([1, 1, 0])
    src = datagen.gen_bytes(kind, n, 9000 + seed)
    if rng.random() < 0.3 and n > 64:
        cut = rng.randrange(1, n); src = (src[cut:] + src[:cut] + src)[:n]
    ok = True
    if n:
        d_src.upload(src)
    clen, crcs = ctx.deflate_raw(d_src, n, chunk, level, last, d_c)
    got = d_c.download(clen).tobytes()
    exp = O.sw_compress("RAW", src, chunk, level, last=last, cap=n * 9 // 8 + 65536 + 64 * (n // chunk + 2))[2]
    ok &= got == exp
    if last and n:                                          # and back
        iu, ol, crc = ctx.inflate_stream(d_c, clen, d_o, chunk)
        ok &= (iu, ol) == (clen, n) and d_o.download(n).tobytes() == src and crc == (zlib.crc32(src) & 0xffffffff)
    if n and seed % 3 == 0:                                 # a foreign stream
        co = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, -15, rng.choice([1, 8, 9]),
                              rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        pieces, pos = [], 0
        while pos < n:
            k = min(n - pos, rng.choice([n, 300000, 65536, 5000]))
            pieces.append(co.compress(src[pos:pos + k])); pos += k
            if pos < n:
                pieces[-1] += co.flush(rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH]))
        comp = b"".join(pieces) + co.flush()
        d_c.upload(comp)
        iu, ol, crc = ctx.inflate_stream(d_c, len(comp), d_o, rng.choice([0, 65536, chunk]))
        ok &= (iu, ol) == (len(comp), n) and d_o.download(n).tobytes() == src
    if not ok:
        bad.append((seed, kind, n, chunk, level, last)); print("MISMATCH", bad[-1], flush=True)
    else:
        n_ok += 1
    seed += 1
print("up to seed %d: %d ok, %d mismatches %s" % (seed - 1, n_ok, len(bad), bad))
