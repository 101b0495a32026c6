#!/usr/bin/env python3
"""Developer tool: bit-exactness at scale - compress <pieces> x <MiB> of DISTINCT data on the GPU and compare every
byte with the CPU oracle (slow: the oracle runs at ~0.11 GB/s).  usage: verify_big.py [pieces] [MiB per piece]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402
import qatzip_amd  # noqa: E402


def main():
    pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n = mb << 20
    ctx = qatzip_amd.Context(0)
    d_src = ctx.alloc(n); d_dst = ctx.alloc(qatzip_amd.max_deflate_len(n, 65536))
    bad = 0
    for i in range(pieces):
        kind = ("silesia", "text", "records", "lzmix")[i % 4] if i % 5 else "silesia"
        src = datagen.gen(kind, n if kind != "lzmix" else min(n, 8 << 20), 9000 + i)
        m = len(src)
        d_src.upload(src)
        t0 = time.perf_counter()
        ctx.deflate_raw_async(d_src, m, 65536, 1, 1, d_dst); ctx.sync()
        got = d_dst.download(ctx.result()).tobytes()
        t1 = time.perf_counter()
        rc, _, exp, _ = O.sw_compress("RAW", src.tobytes(), 65536, 1, cap=m * 9 // 8 + 65536)
        ok = rc == 0 and got == exp
        bad += not ok
        print("piece %d %-8s %4d MiB: gpu %.3f s, oracle %.1f s, %d bytes  %s" %
              (i, kind, m >> 20, t1 - t0, time.perf_counter() - t1, len(got), "identical" if ok else "DIFFERENT"), flush=True)
    print("verify_big: %d of %d pieces differ" % (bad, pieces))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
