#!/usr/bin/env python3
"""Developer tool (CPU model, round 4 K1 experiment a): how many of K1's table lookups a per-wave "this hash key has not
been inserted in this chunk yet" bitmap in LDS would spare.  K1 looks every position of a chunk up (64 at a time); zlib
level 1 inserts parse points and the interiors of matches of at most 4.  For bitmaps of 2^b bits keyed on b bits of the
16-bit hash: the share of lookups whose bit is still clear when the lookup happens (= gathers and far compares skipped).
usage: k1_filter_model.py [chunks]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402

NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CH = 65536
base = datagen.gen("silesia", 128 << 20, 20250523)
rng = np.random.default_rng(5)
picks = rng.choice(len(base) // CH, NCH, replace=False)
tot = 0
skipped = {b: 0 for b in (12, 13, 14, 16)}
first_occ = 0
for p in picks:
    src = base[p * CH:(p + 1) * CH]
    lc, dist = O.deflate_symbols(src.tobytes(), 1)
    s = src.astype(np.uint32)
    h = (((s[:-2] & 0xf) << 12) ^ (s[1:-1] << 6) ^ s[2:]) & 0xffff          # zlib's 3-byte hash at memLevel 9
    n = len(h)
    # inserted positions: every parse point with >= 3 bytes ahead, and the interiors of matches of length <= 4
    ins = np.zeros(n, bool)
    pos = 0
    starts = np.zeros(len(lc), np.int64)
    lens = np.where(dist != 0, lc.astype(np.int64) + 3, 1)
    starts[1:] = np.cumsum(lens)[:-1]
    pp = starts[starts < n]
    ins[pp] = True
    short = (dist != 0) & (lens <= 4)
    for k in (1, 2, 3):
        q = starts[short & (lens > k)] + k
        ins[q[q < n]] = True
    # a lookup at position i sees the insertions of positions < i (K1's window speculation sees them through the exact path)
    order_ins = np.flatnonzero(ins)
    for b in skipped:
        key = h >> (16 - b) if b < 16 else h
        first_ins = np.full(1 << b, n + 1, np.int64)
        np.minimum.at(first_ins, key[order_ins], order_ins)              # position of the key's first insertion
        skipped[b] += int((np.arange(n) <= first_ins[key]).sum())         # lookups at or before it find the bit clear
    tot += n
print("%d chunks of the bench data, %d lookups" % (NCH, tot))
for b, v in skipped.items():
    print("  bitmap of 2^%d bits (%5d B of LDS per wave): %.1f %% of the lookups skipped" % (b, (1 << b) // 8, 100.0 * v / tot))
