#!/usr/bin/env python3
"""Developer tool: sweep K1's residency (QATZIP_AMD_K1_WGS=<persistent workgroups>) and print
the K1 launch time of a full batch plus whole-call compress throughput.  usage: k1_sweep.py [MiB] cfg cfg ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    cfgs = sys.argv[2:] or ["3072"]
    total = mb << 20
    base = datagen.gen("silesia", min(128 << 20, total), 20250523)
    ref = None
    for cfg in cfgs:
        os.environ["QATZIP_AMD_K1_WGS"] = cfg
        ctx = qatzip_amd.Context(0)
        d_src = ctx.alloc(total)
        for off in range(0, total, len(base)):
            d_src.upload(base[:min(len(base), total - off)], off)
        d_dst = ctx.alloc(qatzip_amd.max_deflate_len(total, 65536))
        best = None
        for it in range(3):
            ctx.sync(); t0 = time.perf_counter()
            ctx.deflate_raw_async(d_src, total, 65536, 1, 1, d_dst); ctx.sync()
            dt = time.perf_counter() - t0
            n = ctx.result()
            best = dt if best is None or dt < best else best
        crc = ctx.crc32(d_dst, n)
        if ref is None:
            ref = (n, crc)
        ms = ctx.timing()
        print("cfg %-10s  compress %6.2f GB/s  (first-batch K1 %.2f ms, K2 %.2f ms)  out %d crc %08x %s"
              % (cfg, total / best / 1e9, ms[0], ms[1], n, crc, "OK" if (n, crc) == ref else "MISMATCH"), flush=True)
        d_src.free(); d_dst.free(); ctx.close()


if __name__ == "__main__":
    main()
