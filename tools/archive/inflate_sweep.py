#!/usr/bin/env python3
"""Developer tool: whole-call inflate throughput against QATZIP_AMD_INFLATE / QATZIP_AMD_INFLATE_LPW settings.
usage: inflate_sweep.py [MiB] [lpw ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    lpws = sys.argv[2:] or ["64", "32", "16"]
    n = mb << 20
    base = datagen.gen(os.environ.get("SWEEP_KIND", "silesia"), min(128 << 20, n), 20250523)
    ctx = qatzip_amd.Context(0)
    d_src = ctx.alloc(n)
    for off in range(0, n, len(base)):
        d_src.upload(base[:min(len(base), n - off)], off)
    d_c = ctx.alloc(qatzip_amd.max_deflate_len(n, 65536))
    ctx.deflate_raw_async(d_src, n, 65536, 1, 1, d_c); ctx.sync()
    clen = ctx.result()
    d_o = ctx.alloc(n)
    want = ctx.crc32(d_src, n)
    for lpw in lpws:
        if lpw == "wave":
            os.environ["QATZIP_AMD_INFLATE"] = "wave"
        else:
            os.environ["QATZIP_AMD_INFLATE"] = "lane"; os.environ["QATZIP_AMD_INFLATE_LPW"] = lpw
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            iu, ol, crc = ctx.inflate_stream(d_c, clen, d_o, 65536, want_crc=True)
            best = min(best, time.perf_counter() - t0)
        ms = ctx.inflate_timing()
        print("lpw %-5s inflate %6.2f GB/s  (kernels %.2f ms of which resolve %.2f, crc %.2f ms, wall %.2f ms)  %s" %
              (lpw, n / best / 1e9, ms[0], ms[2], ms[1], best * 1e3, "OK" if (iu, ol, crc) == (clen, n, want) else "MISMATCH"), flush=True)


if __name__ == "__main__":
    main()
