#!/usr/bin/env python3
"""Developer tool (emulator, no GPU): what the K-lanes-per-segment phase A does with the bench data - per segment the trips
of its busiest lane against the trips one lane needs for the whole segment, how the rounds ended, what was handed back.
usage: spec_stats.py [K] [chunks] [kind]   (needs an emulator built with -DQZK_SPEC_STATS: QZSIM_SO)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402
import oracle_lib as O  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NCH = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind = sys.argv[3] if len(sys.argv) > 3 else "silesia"
CH = int(os.environ.get("CHUNK", "65536"))
S = C.CDLL(os.environ.get("QZSIM_SO", "/tmp/w/sim_stats.so"))
S.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
S.sim_spec_stats.restype = C.POINTER(C.c_uint32)
seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
base = datagen.gen(kind, 128 << 20, 20250523) if kind == "silesia" else datagen.gen(kind, NCH * CH * 2, 7)
rng = np.random.default_rng(3)
picks = rng.choice(len(base) // CH, NCH, replace=False)
src = np.concatenate([base[p * CH:(p + 1) * CH] for p in picks]).tobytes()
pieces = [O.deflate_chunk(src[i * CH:(i + 1) * CH], 1, 0 if i + 1 < NCH else 1) for i in range(NCH)]
comp = b"".join(pieces)
segs, off = [], 0
for i, pc in enumerate(pieces):
    segs.append((off, i * CH, len(comp) - off, CH, 0, len(pc))); off += len(pc)
cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); out = np.zeros(NCH * CH + 64, np.uint8)
sa = np.array(segs, dtype=seg_dt); res = np.zeros(NCH, res_dt)
redo = S.sim_inflate_spec(cbuf.ctypes.data, out.ctypes.data, sa.ctypes.data, res.ctypes.data, NCH, K)
assert bytes(out[:NCH * CH]) == src
st = np.ctypeslib.as_array(S.sim_spec_stats(), shape=(1 << 20,))[:NCH * K * 8].reshape(NCH, K, 8).copy()
trips = st & 0x0fffffff; kinds = st >> 28
serial = []
for i in range(NCH):
    lc, dist = O.deflate_symbols(src[i * CH:(i + 1) * CH], 1)
    ism = dist != 0; idx = np.flatnonzero(ism); b = np.concatenate(([-1], idx, [len(lc)])); runs = np.diff(b) - 1
    serial.append(int(np.ceil(runs[0] / 4)) + int(ism.sum()) + int(np.ceil(np.maximum(runs[1:] - 3, 0) / 4).sum()) + len(lc) // 32767 + 1)
serial = np.array(serial)
busiest = trips.sum(2).max(1)          # per segment: the lane with most trips over all rounds (the rounds are wave-synchronous: sum of per-round maxima is the honest figure)
per_round_max = trips.max(1).sum(1)
print("K=%d, %d chunks of %s: handed back %d" % (K, NCH, kind, redo))
print("serial trips  mean %.0f  max %d" % (serial.mean(), serial.max()))
print("K-lane trips (sum over rounds of the busiest lane)  mean %.0f  max %d   speedup mean %.2fx  of the slowest %.2fx" %
      (per_round_max.mean(), per_round_max.max(), serial.mean() / per_round_max.mean(), serial.max() / per_round_max.max()))
order = np.argsort(-serial)
for i in order[:12]:
    print("  seg %3d comp %6d serial %6d -> %6d  rounds:" % (i, len(pieces[i]), serial[i], per_round_max[i]),
          " | ".join(" ".join("%d%s" % (trips[i, j, r], "RSEX"[kinds[i, j, r]]) for j in range(K)) for r in range(8) if trips[i, :, r].any()))

# wave view: the host seats segments by compressed size, sixteen (64 / K) to a wave; a wave's rounds end with their slowest lane
spw = 64 // K
o = np.argsort(-np.array([len(p) >> 5 for p in pieces]), kind="stable")
print("waves of %d segments, largest compressed first: trips = sum over rounds of the slowest lane of the wave" % spw)
for w in range(0, NCH - spw + 1, spw):
    ids = o[w:w + spw]
    wt = trips[ids].max(axis=(0, 1)).sum()
    print("  wave %2d: comp %6d..%6d  serial max %6d  wave trips %6d  (rounds %s)  best case (every segment alone) %6d" %
          (w // spw, len(pieces[ids[-1]]), len(pieces[ids[0]]), serial[ids].max(), wt,
           " ".join(str(int(x)) for x in trips[ids].max(axis=(0, 1)) if x), per_round_max[ids].max()))
