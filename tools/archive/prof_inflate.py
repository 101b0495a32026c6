#!/usr/bin/env python3
"""Developer tool: split K3's time into decode-only (count-only flag) and decode+copy, per data kind."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import qatzip_amd  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "silesia"
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    ctx = qatzip_amd.Context(0)
    base = datagen.gen(kind, min(mb, 64) << 20, 5)
    n = mb << 20
    d_src = ctx.alloc(n)
    for off in range(0, n, base.size):
        d_src.upload(base[:min(base.size, n - off)], off)
    d_c = ctx.alloc(qatzip_amd.max_deflate_len(n))
    clen, _ = ctx.deflate_raw(d_src, n, 65536, 1, 1, d_c, want_crc=False)
    lens = np.zeros(n // 65536, np.uint32)
    ctx.L.qzd_chunk_lens(ctx.h, lens.ctypes.data, len(lens))
    d_o = ctx.alloc(n)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    for flags, name in ((1, "count-only"), (0, "decode+copy")):
        segs = [(int(offs[i]), i * 65536, int(clen - offs[i]), 65536, flags) for i in range(len(lens))]
        ctx.inflate_segments(d_c, d_o, segs)
        t0 = time.perf_counter(); res = ctx.inflate_segments(d_c, d_o, segs); dt = time.perf_counter() - t0
        assert (res["status"] >= 0).all() and (res["out_len"] == 65536).all()
        print("%-12s %-8s %4d MiB: %7.1f ms  %6.2f GB/s  (ratio %.3f)" % (name, kind, mb, dt * 1e3, n / dt / 1e9, clen / n))


if __name__ == "__main__":
    main()
