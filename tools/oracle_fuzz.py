#!/usr/bin/env python3
"""Developer tool: the oracle against the pinned system libz (driven like the reference's call sites, tests/refcalls.py) on
random kind / size / hw_buff_sz / level 1-9 / format / last.  Runs only where zlib 1.2.11 is the system's (the build
container).  usage: oracle_fuzz.py [seconds] [first seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402
import refcalls as R  # noqa: E402

assert R.zlib_pinned(), "needs zlib 1.2.11"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
FM = {"RAW": R.FMT_RAW, "GZIP": R.FMT_GZIP, "GZIP_EXT": R.FMT_GZIP_EXT, "4B": R.FMT_4B, "ZLIB": R.FMT_ZLIB}
t0 = time.time(); n_ok = skipped = 0; bad = []
while time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = rng.choice(datagen.KINDS)
    n = rng.choice([rng.randrange(0, 300), rng.randrange(300, 70000), rng.randrange(70000, 600000)])
    if kind == "lzmix":
        n = min(n, 60000)
    hw = rng.choice([1024, 4096, 16384, 65536, 131072, 524288]); lvl = rng.randrange(1, 10)
    fmt = rng.choice(list(FM)); last = rng.choice([1, 1, 0])
    src = datagen.gen_bytes(kind, n, 70000 + seed)
    if rng.random() < 0.3 and n > 64:
        cut = rng.randrange(1, n); src = (src[cut:] + src[:cut] + src)[:n]
    seed += 1
    if fmt == "4B" and last == 0:       # the reference reports a length without the 4 reserved bytes there (src/qatzip_sw.c:167-170
        skipped += 1; continue          # against :242-244): the oracle restates that, the model in refcalls.py does not
    a = O.sw_compress(fmt, src, hw, lvl, last=last, cap=n * 9 // 8 + 65536 + 64 * (n // hw + 2))[2]
    if a != R.sw_compress(FM[fmt], src, hw, lvl, last):
        bad.append((seed - 1, kind, n, hw, lvl, fmt, last)); print("MISMATCH", bad[-1], flush=True)
    else:
        n_ok += 1
print("oracle vs libz %s: %d ok, %d mismatches, %d skipped" % ("1.2.11", n_ok, len(bad), skipped))
