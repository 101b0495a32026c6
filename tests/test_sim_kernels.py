"""CPU-side fuzzing of the HIP kernel bodies through the SIMT emulator (tests/sim/): the same
source the GPU runs, executed lane-by-lane on fibers, against the oracle.  Keeps kernel logic
testable in the build container (no GPU here); the -m gpu tests are the parity tests proper."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

import datagen
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
SIMDIR = os.path.join(HERE, "sim")
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(SIMDIR, "libqzsim.so")
    deps = [os.path.join(SIMDIR, f) for f in ("sim_driver.cpp", "hipsim.h")]
    csrc = os.path.join(ROOT, "qatzip_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-I", SIMDIR,
                               "-Wno-unused-function", "-o", so, os.path.join(SIMDIR, "sim_driver.cpp")])
    S = C.CDLL(so)
    S.sim_lz77.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    S.sim_deflate.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_uint64),
                              C.c_void_p]
    S.sim_deflate_fused.argtypes = S.sim_deflate.argtypes
    S.sim_deflate_wide.argtypes = S.sim_deflate.argtypes
    return S


def _sim_deflate(S, src, chunk, last=1, fused=False):
    n = len(src)
    nch = max(1, (n + chunk - 1) // chunk)
    cap = n * 9 // 8 + 4096 * (nch + 1)
    out = C.create_string_buffer(cap); ol = C.c_uint64(0)
    crcs = np.zeros(nch, np.uint32)
    (S.sim_deflate_fused if fused else S.sim_deflate)(src, n, chunk, last, out, C.byref(ol), crcs.ctypes.data)
    return out.raw[:ol.value], crcs


def test_lz77_symbols_match_oracle(sim):
    for kind in datagen.KINDS:
        for n in (1, 2, 3, 17, 300, 4000, 30000):
            src = datagen.gen_bytes(kind, n, 31)
            lc = np.zeros(n + 64, np.uint8); dist = np.zeros(n + 64, np.uint16)
            meta = np.zeros(sim.sim_meta_size() // 4, np.uint32)
            sim.sim_lz77(src, n, 65536, lc.ctypes.data, dist.ctypes.data, meta.ctypes.data)
            olc, odist = O.deflate_symbols(src, 1)
            ns = int(meta[0])
            assert ns == len(olc), (kind, n)
            assert (lc[:ns] == olc).all() and (dist[:ns] == odist).all(), (kind, n)


@pytest.mark.parametrize("kind", datagen.KINDS)
def test_deflate_stream_matches_oracle(sim, kind):
    for n, chunk in ((0, 65536), (5, 1024), (65536, 65536), (65400, 65536), (70000, 16384), (140000, 131072)):
        if kind == "lzmix" and n > 70000:
            n = 66000
        src = datagen.gen_bytes(kind, n, 5)
        out, crcs = _sim_deflate(sim, src, chunk)
        rc, _, exp, _ = O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 8192)
        assert out == exp, (kind, n, chunk)
        for i in range(len(crcs)):
            assert crcs[i] == (zlib.crc32(src[i * chunk:(i + 1) * chunk]) & 0xffffffff)


def test_deflate_fused_k1_k2(sim):
    """The product's launch shape: the wave that parsed a chunk also codes it (K2 inside K1's pull loop, in the LDS the
    parse no longer needs), chunk after chunk on the same wave."""
    for kind, n, chunk in (("silesia", 200000, 65536), ("lzmix", 66000, 16384), ("runs", 140000, 131072),
                           ("text", 9000, 1024), ("allA", 70000, 65536), ("rand", 0, 65536), ("rand", 70000, 16384),
                           ("mod200", 300000, 65536), ("text", 100000, 4096), ("text", 70001, 65536), ("silesia", 65539, 65536),
                           ("records", 131074, 65536), ("text", 1027, 1024), ("text", 3, 65536), ("text", 259, 65536)):
        src = datagen.gen_bytes(kind, n, 77)
        for last in (1, 0):
            out, crcs = _sim_deflate(sim, src, chunk, last=last, fused=True)
            assert out == O.sw_compress("RAW", src, chunk, 1, last=last, cap=n * 9 // 8 + 8192)[2], (kind, n, chunk, last)
            for i in range(len(crcs) if n else 0):      # the CRC that rides along K1's input reads
                assert crcs[i] == (zlib.crc32(src[i * chunk:(i + 1) * chunk]) & 0xffffffff), (kind, n, chunk, i)


def test_deflate_not_last(sim):
    src = datagen.gen_bytes("text", 40000, 2)
    out, _ = _sim_deflate(sim, src, 16384, last=0)
    assert out == O.sw_compress("RAW", src, 16384, 1, last=0)[2]


def test_lane_kernels_match_oracle(sim):
    """K1b (one chunk per lane) and K3b (one segment per lane): same contracts, opposite mapping."""
    sim.sim_deflate_lane.argtypes = sim.sim_deflate.argtypes
    sim.sim_inflate_lane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
    for kind in datagen.KINDS:
        for n, chunk in ((0, 65536), (70000, 16384), (65400, 65536), (140000, 131072), (9000, 1024)):
            if kind == "lzmix" and n > 70000:
                n = 66000
            src = datagen.gen_bytes(kind, n, 8)
            nch = max(1, (n + chunk - 1) // chunk)
            cap = n * 9 // 8 + 4096 * (nch + 1)
            out = C.create_string_buffer(cap); ol = C.c_uint64(0); crcs = np.zeros(nch, np.uint32)
            sim.sim_deflate_lane(src, n, chunk, 1, out, C.byref(ol), crcs.ctypes.data)
            comp = out.raw[:ol.value]
            assert comp == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 8192)[2], (kind, n, chunk)
            # decode every chunk's segment with the lane inflate kernel
            import refcalls as R
            pieces = R.raw_chunks(src, chunk, 1, 1) if zlib.ZLIB_RUNTIME_VERSION == "1.2.11" else None
            if pieces is None or n == 0:
                continue
            segs, off, oo = [], 0, 0
            for pc in pieces:
                ln = min(chunk, n - oo)
                segs.append((off, oo, len(comp) - off, ln, 0, 0)); off += len(pc); oo += ln
            cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.zeros(n + 64, np.uint8)
            sa = np.array(segs, dtype=seg_dt); res = np.zeros(len(segs), res_dt)
            sim.sim_inflate_lane(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs))
            assert bytes(obuf[:n]) == src and (res["status"] >= 0).all(), (kind, n, chunk)
            assert [int(r["in_used"]) for r in res] == [len(pc) for pc in pieces]


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6, 7, 8, 9])
def test_every_zlib_level_matches_oracle(sim, level):
    """The general parse kernel (zlib's own loop per lane) at the greedy levels 2-3 and the lazy levels 4-9: the bytes
    the software path writes at that comp_lvl (oracle pinned to libz 1.2.11 at every level, tests/test_oracle.py)."""
    sim.sim_deflate_level.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    cases = [("text", 70000, 16384), ("silesia", 66000, 65536), ("runs", 140000, 131072), ("rand", 9000, 1024),
             ("lzmix", 30000, 65536), ("records", 200000, 524288), ("allA", 70000, 65536), ("mod200", 40000, 65536), ("text", 0, 65536), ("text", 2, 1024)]
    for kind, n, chunk in cases:
        src = datagen.gen_bytes(kind, n, 20 + level)
        nch = max(1, (n + chunk - 1) // chunk)
        cap = n * 9 // 8 + 4096 * (nch + 1)
        for last in (1, 0):
            out = C.create_string_buffer(cap); ol = C.c_uint64(0); crcs = np.zeros(nch, np.uint32)
            sim.sim_deflate_level(src, n, chunk, last, level, out, C.byref(ol), crcs.ctypes.data)
            assert out.raw[:ol.value] == O.sw_compress("RAW", src, chunk, level, last=last, cap=cap)[2], (kind, n, chunk, level, last)


@pytest.mark.parametrize("level", [4, 5, 6, 7, 8, 9])
def test_lazy_levels_by_parallel_search_match_oracle(sim, level):
    """comp_lvl 4-9 through the three lazy kernels: chains built once, every position's search done by its own lane
    (full and quarter chain), then the serial lazy parse over the stored answers - the same bytes as zlib's
    deflate_slow, including the window slides of chunks above 64 KB and streams that stay open (last = 0)."""
    sim.sim_deflate_lazy.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    # (the emulator meets at every hop of the search, so sizes are kept small; the GPU tests run the large ones)
    cases = [("text", 20000, 16384), ("silesia", 30000, 65536), ("rand", 5000, 1024), ("lzmix", 12000, 65536),
             ("allA", 70000, 65536), ("text", 0, 65536), ("text", 2, 1024), ("text", 3, 1024),
             ("runs", 65536 + 300, 131072), ("records", 98304 + 7, 131072)]
    if level in (5, 7, 8):
        cases = cases[:2] + cases[-1:]
    for kind, n, chunk in cases:
        src = datagen.gen_bytes(kind, n, 30 + level)
        nch = max(1, (n + chunk - 1) // chunk)
        cap = n * 9 // 8 + 4096 * (nch + 1)
        for last in (1, 0):
            out = C.create_string_buffer(cap); ol = C.c_uint64(0); crcs = np.zeros(nch, np.uint32)
            sim.sim_deflate_lazy(src, n, chunk, last, level, out, C.byref(ol), crcs.ctypes.data)
            assert out.raw[:ol.value] == O.sw_compress("RAW", src, chunk, level, last=last, cap=cap)[2], (kind, n, chunk, level, last)


@pytest.mark.parametrize("level", [1, 3, 6])
def test_coalesced_launch_of_small_requests(sim, level):
    """Many small requests in one launch: every request starts on a chunk boundary, chunks carry their own length and
    'closes the stream' flag.  Each request's bytes must be what a call of its own produces."""
    chunk = 16384
    reqs = [datagen.gen_bytes(k, n, 40 + i) for i, (k, n) in enumerate(
        (("text", 16384), ("silesia", 5000), ("rand", 0), ("runs", 40000), ("text", 1), ("lzmix", 16385), ("records", 32768), ("allA", 3)))]
    slots, cdesc = [], []
    for r in reqs:
        nch = max(1, (len(r) + chunk - 1) // chunk)
        for k in range(nch):
            piece = r[k * chunk:(k + 1) * chunk]
            slots.append(piece + bytes([0xEE]) * (chunk - len(piece)))          # the padding is never part of the result
            cdesc.append(len(piece) | (0x80000000 if k == nch - 1 else 0))
    buf = b"".join(slots); nch = len(cdesc)
    cd = np.array(cdesc, np.uint32); lens = np.zeros(nch, np.uint32); crcs = np.zeros(nch, np.uint32)
    out = C.create_string_buffer(len(buf) * 9 // 8 + 4096 * nch)
    sim.sim_deflate_ragged.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    total = sim.sim_deflate_ragged(buf, nch, chunk, cd.ctypes.data, level, out, lens.ctypes.data, crcs.ctypes.data)
    pos = k = 0
    for r in reqs:
        n = max(1, (len(r) + chunk - 1) // chunk)
        ln = int(lens[k:k + n].sum())
        assert out.raw[pos:pos + ln] == O.sw_compress("RAW", r, chunk, level)[2], (len(r), level)
        assert [int(c) for c in crcs[k:k + n]] == [zlib.crc32(r[j * chunk:(j + 1) * chunk]) & 0xffffffff for j in range(n)]
        pos += ln; k += n
    assert pos == total


@pytest.mark.parametrize("variant", ["sim_inflate"])
def test_wave_inflate_kernels(sim, variant):
    """K3, one wave per segment (literal runs decoded by bit-offset speculation): flush-marker segments of our own streams, and whole foreign streams decoded straight through (level 9 text with 32 KiB distances, stored and
    fixed blocks, sync-flush history), plus the count-only pass and the error codes."""
    fn = getattr(sim, variant)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])

    def run(comp, segs, n):
        cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.full(n + 64, 0xAA, np.uint8)
        sa = np.array(segs, dtype=seg_dt); res = np.zeros(len(segs), res_dt)
        fn(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs))
        return bytes(obuf[:n]), res, bytes(obuf[n:n + 64])

    # 1. our own chunked streams, one segment per chunk
    for kind, n, chunk in (("text", 70000, 16384), ("silesia", 140000, 65536), ("rand", 40000, 16384), ("runs", 9000, 1024), ("allA", 70000, 65536)):
        src = datagen.gen_bytes(kind, n, 9)
        comp = O.sw_compress("RAW", src, chunk, 1)[2]
        segs, off, oo = [], 0, 0
        for k in range((n + chunk - 1) // chunk):
            ln = min(chunk, n - oo)
            plen = len(O.sw_compress("RAW", src[oo:oo + ln], chunk, 1, last=1 if oo + ln == n else 0)[2])
            segs.append((off, oo, len(comp) - off, ln, 0, 0)); off += plen; oo += ln
        got, res, tail = run(comp, segs, n)
        assert got == src and (res["status"] >= 0).all() and tail == b"\xaa" * 64, (variant, kind)
        cnt = [(a, b, c, d, 1, 0) for a, b, c, d, _, _ in segs]          # count only: sizes, nothing written
        got2, res2, _ = run(comp, cnt, n)
        assert got2 == b"\xaa" * n and [int(x) for x in res2["out_len"]] == [s_[3] for s_ in segs]
    # 2. foreign streams straight through (flag 2): long distances, every block type, Z_SYNC_FLUSH in the middle
    text = datagen.gen_bytes("text", 150000, 3)
    co = zlib.compressobj(9, zlib.DEFLATED, -15)
    a = co.compress(text[:90000]) + co.flush(zlib.Z_SYNC_FLUSH)
    bpart = co.compress(text[90000:] + datagen.gen_bytes("rand", 70000, 4) + text[:40000]) + co.flush()
    whole = text + datagen.gen_bytes("rand", 70000, 4) + text[:40000]
    got, res, tail = run(a + bpart, [(0, 0, len(a + bpart), len(whole), 2, 0)], len(whole))
    assert got == whole and res[0]["status"] == 0 and res[0]["in_used"] == len(a + bpart) and tail == b"\xaa" * 64
    fixed = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    fx = fixed.compress(text[:30000]) + fixed.flush()
    got, res, _ = run(fx, [(0, 0, len(fx), 30000, 2, 0)], 30000)
    assert got == text[:30000] and res[0]["status"] == 0
    # 3. errors: output too small, truncated input, damaged data
    got, res, tail = run(fx, [(0, 0, len(fx), 29000, 2, 0)], 29000)
    assert res[0]["status"] == -2 and tail == b"\xaa" * 64
    got, res, _ = run(fx[:len(fx) // 2], [(0, 0, len(fx) // 2, 30000, 2, 0)], 30000)
    assert res[0]["status"] in (-3, -1)
    bad = bytearray(a + bpart); bad[1000] ^= 0x40
    got, res, _ = run(bytes(bad), [(0, 0, len(bad), len(whole), 2, 0)], len(whole))
    assert res[0]["status"] < 0 or got != whole


def test_adler_chunks_kernel(sim):
    """the DEFLATE_ZLIB trailer checksum: per-chunk Adler-32 on the emulator against zlib.adler32"""
    sim.sim_adler.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p]
    for kind, n, chunk in (("rand", 0, 65536), ("rand", 1, 65536), ("text", 255, 1024), ("silesia", 200001, 65536),
                           ("allA", 524288, 524288), ("rand", 70000, 16384)):
        src = datagen.gen_bytes(kind, n, 4) if kind != "allA" else b"\xff" * n
        nch = max(1, (n + chunk - 1) // chunk)
        out = np.zeros(nch, np.uint32)
        sim.sim_adler(src, n, chunk, out.ctypes.data)
        for i in range(nch):
            assert out[i] == zlib.adler32(src[i * chunk:(i + 1) * chunk]), (kind, n, chunk, i)


def test_speculative_inflate_matches_serial(sim):
    """K3b with K lanes per segment (sub-segment speculation, qzk_inflate_spec.h): same bytes and same per-segment
    results as the one-lane-per-segment phase A; segments it cannot take (stored / several blocks) come back through
    the serial kernel"""
    import refcalls as R
    if not R.zlib_pinned():
        pytest.skip("needs the pinned zlib to cut a stream into its per-chunk pieces")
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
    sim.sim_inflate_lane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    sim.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    taken = 0
    for kind, n, chunk in (("silesia", 400000, 65536), ("text", 300000, 65536), ("records", 200000, 32768),
                           ("runs", 140000, 65536), ("allA", 70000, 65536), ("rand", 140000, 65536),
                           ("lzmix", 66000, 65536), ("mod200", 100000, 16384), ("text", 9000, 1024)):
        src = datagen.gen_bytes(kind, n, 12)
        pieces = R.raw_chunks(src, chunk, 1, 1)
        comp = b"".join(pieces)
        segs, off, oo = [], 0, 0
        for i, pc in enumerate(pieces):
            ln = min(chunk, n - oo)
            # like the optimistic pass of the host: in_len runs to the end of the stream, pad carries the real length
            segs.append((off, oo, len(comp) - off, ln, 0, len(pc))); off += len(pc); oo += ln
        cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy()
        sa = np.array(segs, dtype=seg_dt)
        ref_out = np.zeros(n + 64, np.uint8); ref_res = np.zeros(len(segs), res_dt)
        sim.sim_inflate_lane(cbuf.ctypes.data, ref_out.ctypes.data, sa.ctypes.data, ref_res.ctypes.data, len(segs))
        assert bytes(ref_out[:n]) == src
        for K in (2, 4, 8, 16, 32):
            out = np.zeros(n + 64, np.uint8); res = np.zeros(len(segs), res_dt)
            redone = sim.sim_inflate_spec(cbuf.ctypes.data, out.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs), K)
            assert bytes(out[:n]) == src, (kind, n, chunk, K)
            for f in ("status", "in_used", "out_len"):
                assert (res[f] == ref_res[f]).all(), (kind, chunk, K, f, res[f], ref_res[f])
            taken += len(segs) - redone
    assert taken > 50            # the speculative kernel really decoded most of the ordinary segments itself


def test_speculative_inflate_where_lanes_do_not_fall_into_step(sim):
    """codes of one length (bytes drawn evenly from 64 or 256 values: a decoder that starts off a symbol boundary stays off
    it) and of nearly one length (48 values), alone and between text: the lanes of a group run their bounded reach without
    falling into step, give the rest back, take their unused pieces back - the bytes are zlib's all the same, through
    the continuation rounds or through the serial kernel (qzk_inflate_spec.h, QZK_SPEC_REACH)"""
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
    sim.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    rng = np.random.default_rng(77)
    text = datagen.gen_bytes("text", 40000, 3)
    flat64 = bytes(rng.integers(0, 64, 65536, dtype=np.uint8))
    flat48 = bytes(rng.integers(0, 48, 65536, dtype=np.uint8))
    flat256 = bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    chunks = [flat64, flat48, text[:20000] + flat64[:30000] + text[20000:35000], flat48[:25000] + text + flat48[25000:],
              flat256 + text[:9000] + flat64[:20000]]
    for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY):
        co = zlib.compressobj(1, zlib.DEFLATED, -15, 8, strategy)
        pieces = [co.compress(ch) + co.flush(zlib.Z_FULL_FLUSH) for ch in chunks[:-1]]
        pieces.append(co.compress(chunks[-1]) + co.flush())
        comp = b"".join(pieces); src = b"".join(chunks); n = len(src)
        segs, off, oo = [], 0, 0
        for pc, ch in zip(pieces, chunks):
            segs.append((off, oo, len(comp) - off, len(ch), 0, len(pc))); off += len(pc); oo += len(ch)
        cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy()
        sa = np.array(segs, dtype=seg_dt)
        for K in (4, 8, 16, 32):
            out = np.full(n + 64, 0xAA, np.uint8); res = np.zeros(len(segs), res_dt)
            sim.sim_inflate_spec(cbuf.ctypes.data, out.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs), K)
            assert bytes(out[:n]) == src and bytes(out[n:]) == b"\xaa" * 64, (strategy, K)
            assert (res["status"] >= 0).all() and [int(r["in_used"]) for r in res] == [len(pc) for pc in pieces], (strategy, K, res)


def test_a_match_before_the_segment_found_by_phase_b_is_an_error(sim):
    """a distance made too large by a flipped bit, in a part of the block that a lane other than lane 0 decodes: phase A
    cannot know (that lane does not know where in the output it stands), phase B must say so - and its answer must survive
    the batches that follow (round 5's phase B overwrote a batch's error with the next batch's 0: tools/sim_fuzz_corrupt.py,
    seed 188949; zlib: "invalid distance too far back", src/qatzip_sw.c:353-359 makes that QZ_DATA_ERROR)"""
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
    sim.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    n = 14012
    src = datagen.gen_bytes("runs", n, 9000 + 188949)
    co = zlib.compressobj(1, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)      # (fixed codes: a flipped bit in a distance's extra bits stays a distance)
    good = co.compress(src) + co.flush()
    too_far = accepted = 0
    for at in range(len(good) * 55 // 100, len(good) - 1):
        for bit in range(8):
            comp = bytearray(good); comp[at] ^= 1 << bit; comp = bytes(comp)
            want, why = None, ""
            try:
                d = zlib.decompressobj(-15)
                o = d.decompress(comp, n + 1)
                if d.eof and len(o) == n:
                    want = o
            except zlib.error as e:
                why = str(e)
            if "too far back" not in why:
                continue
            for K in (4, 8):
                cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.full(n + 64, 0xAA, np.uint8)
                sa = np.array([(0, 0, len(comp), n, 0, len(comp))], dtype=seg_dt); res = np.zeros(1, res_dt)
                sim.sim_inflate_spec(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, 1, K)
                assert bytes(obuf[n:]) == b"\xaa" * 64, (at, bit, K)
                if res[0]["status"] >= 0 and res[0]["out_len"] == n:
                    assert want is not None and bytes(obuf[:n]) == want, (at, bit, K, why)
                    accepted += 1
                elif "too far back" in why:
                    too_far += 1
    assert too_far > 500 and accepted == 0, (too_far, accepted)


def test_damaged_streams_end_in_an_error_or_in_zlibs_own_bytes(sim):
    """a decoder must never crash, hang or write outside its output: a segment with flipped bits (most of them in the block
    header, which is decoded in LDS and registers since round 4) or cut short is reported as an error - or, when zlib decodes
    the damaged stream too and it fills the segment exactly, comes out as the bytes zlib produces (tools/sim_fuzz_corrupt.py
    is the campaign; the reference's flipped-byte cases: SURVEY 8b, src/qatzip_sw.c:357-361)"""
    import random
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
    for f in (sim.sim_inflate, sim.sim_inflate_lane):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    sim.sim_inflate_spec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    decoders = (("wave", sim.sim_inflate), ("lane", sim.sim_inflate_lane),
                ("k4", lambda a, b, c, d, e: sim.sim_inflate_spec(a, b, c, d, e, 4)),
                ("k16", lambda a, b, c, d, e: sim.sim_inflate_spec(a, b, c, d, e, 16)))
    errors = same = 0
    for seed in range(1, 241):
        rng = random.Random(seed)
        kind = rng.choice(datagen.KINDS)
        n = rng.choice([rng.randrange(1, 300), rng.randrange(300, 20000)])
        src = datagen.gen_bytes(kind, n, 9000 + seed)
        co = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15, 9, rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY]))
        comp = bytearray(co.compress(src) + co.flush())
        if rng.random() < 0.15 and len(comp) > 4:
            comp = comp[:rng.randrange(1, len(comp))]
        else:
            for _ in range(rng.choice([1, 1, 2, 5])):
                at = rng.randrange(0, min(len(comp), 120)) if rng.random() < 0.7 else rng.randrange(0, len(comp))
                comp[at] ^= 1 << rng.randrange(8)
        comp = bytes(comp)
        want = None
        try:
            d = zlib.decompressobj(-15)
            out = d.decompress(comp, n + 1)
            if d.eof and len(out) == n:
                want = out
        except zlib.error:
            pass
        for name, fn in decoders:
            cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.full(n + 64, 0xAA, np.uint8)
            sa = np.array([(0, 0, len(comp), n, 0, len(comp))], dtype=seg_dt); res = np.zeros(1, res_dt)
            fn(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, 1)
            assert bytes(obuf[n:]) == b"\xaa" * 64, (seed, name)
            if res[0]["status"] >= 0 and res[0]["out_len"] == n:
                assert want is not None and bytes(obuf[:n]) == want, (seed, name, kind, n)
                same += 1
            else:
                errors += 1
    assert errors > 200 and same > 50, (errors, same)


def test_lz4_linked_frames_match_liblz4_goldens(sim):
    """a QZ_LZ4 call above 64 KB: ONE frame with linked blocks (LZ4F_compressFrame, src/qatzip_sw.c:451-456) - the kernel
    against liblz4 1.9.3's own frames (tests/golden/lz4_linked) and the oracle on a few more shapes"""
    import json
    sim.sim_lz4c_linked.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
    d = os.path.join(HERE, "golden", "lz4_linked")
    with open(os.path.join(d, "index.json")) as f:
        idx = json.load(f)
    cases = [(fr["kind"], fr["n"], fr["seed"], open(os.path.join(d, fr["file"]), "rb").read()) for fr in idx["frames"]]
    for kind, n, seed in (("text", 65536 + 12, 3), ("text", 65536 + 13, 4), ("silesia", 3 * 65536, 5), ("records", 131072 + 5, 6)):
        src = datagen.gen_bytes(kind, n, seed)
        cases.append((kind, n, seed, O.sw_compress("LZ4", src, 65536, 1, cap=n + n // 255 + 1000)[2]))
    # an incompressible block between compressible ones: stored, and the table keeps what the attempt inserted
    mix = datagen.gen_bytes("text", 70000, 8) + datagen.gen_bytes("rand", 66000, 9) + datagen.gen_bytes("text", 70000, 8)
    cases.append(("mix", len(mix), -1, O.sw_compress("LZ4", mix, 65536, 1, cap=len(mix) + 4000)[2]))
    for kind, n, seed, exp in cases:
        src = mix if kind == "mix" else datagen.gen_bytes(kind, n, seed)
        out = C.create_string_buffer(n + n // 255 + 4096); ol = C.c_uint32(0)
        sim.sim_lz4c_linked(src, n, out, C.byref(ol))
        assert out.raw[:ol.value] == exp, (kind, n, seed, ol.value, len(exp))
    # the hardware framing's chunks above 64 KB: a frame per chunk, all in one launch (compress_lz4_hw) - each slot must hold
    # what the one-frame kernel writes for that chunk alone
    sim.sim_lz4c_linked_many.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    src = datagen.gen_bytes("silesia", 3 * 131072 + 70000, 11)
    chunk, nfr = 131072, 4
    stride = (chunk + 15 + 4 * 2 + 8 + 64 + 15) & ~15
    slots = C.create_string_buffer(nfr * stride); lens = (C.c_uint32 * nfr)()
    sim.sim_lz4c_linked_many(src, len(src), chunk, nfr, slots, stride, lens)
    for k in range(nfr):
        part = src[k * chunk:(k + 1) * chunk]
        out = C.create_string_buffer(len(part) + 4096); ol = C.c_uint32(0)
        sim.sim_lz4c_linked(part, len(part), out, C.byref(ol))
        assert slots.raw[k * stride:k * stride + lens[k]] == out.raw[:ol.value], k


def test_lz4_frames_match_oracle(sim):
    """K4: one frame per call of at most 64 KB (LZ4F_compressFrame with one independent block, src/qatzip_sw.c:443-471):
    the kernel's 16-bit hash table and its elected last writer against the oracle, every kind, sizes around the limits of
    lz4's loop (MFLIMIT, 64 KB) and incompressible input (stored block)"""
    sim.sim_lz4c.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    sim.sim_lz4c_pull.argtypes = sim.sim_lz4c.argtypes + [C.c_uint32]
    for kind in datagen.KINDS:
        for n, fs in ((0, 65536), (1, 65536), (12, 65536), (13, 65536), (14, 65536), (300, 65536), (65535, 65536), (65536, 65536),
                      (200000, 65536), (70000, 16384), (9000, 1024)):
            if kind == "lzmix" and n > 70000:
                n = 66000
            src = datagen.gen_bytes(kind, n, 41)
            nfr = max(1, (n + fs - 1) // fs)
            stride = (fs + 15 + 4 + 8 + 64 + 15) & ~15
            slots = np.zeros(nfr * stride, np.uint8); lens = np.zeros(nfr, np.uint32)
            sim.sim_lz4c(src, n, fs, slots.ctypes.data, stride, lens.ctypes.data)
            exps = []
            for i in range(nfr):
                piece = src[i * fs:(i + 1) * fs]
                exp = O.sw_compress("LZ4", piece, 65536, 1, cap=len(piece) + len(piece) // 255 + 200)[2]
                exps.append(exp)
                got = bytes(slots[i * stride:i * stride + int(lens[i])])
                assert got == exp, (kind, n, fs, i, int(lens[i]), len(exp))
            if nfr > 1 or n in (0, 300, 65536):
                # the persistent form (qzk_lz4c_pull_kernel): two waves pull the frames, their tables live outside LDS
                slots[:] = 0; lens[:] = 0
                sim.sim_lz4c_pull(src, n, fs, slots.ctypes.data, stride, lens.ctypes.data, 2)
                for i in range(nfr):
                    assert bytes(slots[i * stride:i * stride + int(lens[i])]) == exps[i], (kind, n, fs, i, "pull")


def test_wide_window_parse_is_exact(sim):
    """K1w (experimental, qzk_deflate_wide.h): one chunk of at most 64 KB per 1024-thread workgroup, the parse found as the
    fixpoint of assume-inserted / match / parse rounds over 1024-position windows, zlib's own head[] / prev[] on chip -
    the same bytes as the software path on every kind, at the sizes where zlib's end-of-chunk rules bite"""
    for kind in datagen.KINDS:
        for n, chunk in ((0, 65536), (1, 65536), (3, 65536), (300, 65536), (1020, 65536), (1024, 65536), (1025, 65536), (9000, 1024),
                         (65274, 65536), (65275, 65536), (65400, 65536), (65536, 65536), (200000, 65536), (70000, 16384)):
            if kind == "lzmix" and n > 70000:
                n = 66000
            src = datagen.gen_bytes(kind, n, 53)
            nch = max(1, (n + chunk - 1) // chunk)
            cap = n * 9 // 8 + 4096 * (nch + 1)
            for last in (1, 0):
                out = C.create_string_buffer(cap); ol = C.c_uint64(0); crcs = np.zeros(nch, np.uint32)
                sim.sim_deflate_wide(src, n, chunk, last, out, C.byref(ol), crcs.ctypes.data)
                assert out.raw[:ol.value] == O.sw_compress("RAW", src, chunk, 1, last=last, cap=cap)[2], (kind, n, chunk, last)


def _fixed_block_bits():
    """a tiny fixed-Huffman DEFLATE writer (RFC 1951 3.2.6) for hand-built streams"""
    class W:
        def __init__(self):
            self.acc = 0; self.n = 0; self.out = bytearray()
        def bits(self, v, k):                      # LSB first (header fields, extra bits)
            self.acc |= v << self.n; self.n += k
            while self.n >= 8:
                self.out.append(self.acc & 0xff); self.acc >>= 8; self.n -= 8
        def code(self, c, k):                      # Huffman codes go MSB first
            self.bits(int(format(c, "0%db" % k)[::-1], 2), k)
        def lit(self, b):
            self.code(0x30 + b, 8) if b < 144 else self.code(0x190 + b - 144, 9)
        def eob(self):
            self.code(0, 7)
        def match3(self, dist_code):               # length 3 = symbol 257 (7 bits), 5-bit distance code, no extra bits
            self.code(1, 7); self.code(dist_code, 5)
        def align(self):
            if self.n: self.bits(0, 8 - self.n)
    return W()


def test_lane_inflate_stored_block_between_literals_and_matches(sim):
    """Advisor finding (round 2, high): literals, then a stored block, then eight or more matches in one round of phase A left
    nine sequences for eight staging slots - the last one was dropped and the output was wrong with status 0."""
    sim.sim_inflate_lane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    seg_dt = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
    res_dt = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
    for nlit, nstored, nmatch in ((400, 100, 40), (1, 100, 8), (5, 7, 9), (37, 65535, 64), (3, 1, 8)):
        w = _fixed_block_bits()
        w.bits(0, 1); w.bits(1, 2)
        for i in range(nlit):
            w.lit(97 + i % 23)
        w.eob()
        w.bits(0, 1); w.bits(0, 2); w.align()
        w.bits(nstored, 16); w.bits(nstored ^ 0xffff, 16)
        w.out += bytes((i * 7 + 3) & 0xff for i in range(nstored))
        w.bits(1, 1); w.bits(1, 2)
        for i in range(nmatch):
            w.match3(i % 4)
        w.eob(); w.align()
        comp = bytes(w.out)
        exp = zlib.decompressobj(-15).decompress(comp)
        assert len(exp) == nlit + nstored + 3 * nmatch
        cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy(); obuf = np.zeros(len(exp) + 64, np.uint8)
        sa = np.array([(0, 0, len(comp), len(exp), 0, 0)], dtype=seg_dt); res = np.zeros(1, res_dt)
        sim.sim_inflate_lane(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, 1)
        assert int(res["status"][0]) >= 0 and int(res["out_len"][0]) == len(exp), (nlit, nstored, nmatch)
        assert bytes(obuf[:len(exp)]) == exp, (nlit, nstored, nmatch)


def _lz4_hw_frame(piece, sw_frame):
    """what the reference's hardware path puts around a chunk (src/qatzip_lz4.c:104-143): qzLZ4HeaderGen's 15 bytes - magic,
    FLG 0x4C, BD 64 KB, content size = the chunk's bytes, (XXH32(FLG..size) >> 8) & 0xff - then the block(s), then
    qzLZ4FooterGen's end mark + XXH32 of the chunk; the part behind the header is the software frame's"""
    desc = bytes([0x4C, 0x40]) + len(piece).to_bytes(8, "little")
    return bytes([0x04, 0x22, 0x4D, 0x18]) + desc + bytes([(O.lib().qzo_xxh32(desc, len(desc), 0) >> 8) & 0xff]) + sw_frame[15:]


def test_lz4_hardware_path_header(sim):
    """K4 with hw_hdr: a frame per chunk behind the header qzLZ4HeaderGen writes (FLG 0x4C, content size, header checksum),
    expectation built from the reference's generator + the oracle's frame body"""
    sim.sim_lz4c_hw.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    for kind in ("silesia", "rand", "runs", "allA"):
        for n, fs in ((65536, 65536), (200000, 65536), (70000, 16384), (1024, 65536)):
            src = datagen.gen_bytes(kind, n, 43)
            nfr = max(1, (n + fs - 1) // fs)
            stride = (fs + 15 + 4 + 8 + 64 + 15) & ~15
            slots = np.zeros(nfr * stride, np.uint8); lens = np.zeros(nfr, np.uint32)
            sim.sim_lz4c_hw(src, n, fs, slots.ctypes.data, stride, lens.ctypes.data)
            for i in range(nfr):
                piece = src[i * fs:(i + 1) * fs]
                sw = O.sw_compress("LZ4", piece, 65536, 1, cap=len(piece) + len(piece) // 255 + 200)[2]
                got = bytes(slots[i * stride:i * stride + int(lens[i])])
                assert got == _lz4_hw_frame(piece, sw), (kind, n, fs, i)


LZ4SEG_DT = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4")])
LZ4RES_DT = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("pad", "<u4")])


def _sim_lz4d(sim, frames, caps, lead=0, guard=64):
    """decode `frames` (one segment each) into one buffer, outputs back to back behind `lead` bytes; returns (bytes per
    segment, results, the guard bytes around the outputs)"""
    sim.sim_lz4d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    comp = b"".join(frames)
    cbuf = np.frombuffer(comp + b"\0" * 64, np.uint8).copy()
    total = lead + sum(caps)
    obuf = np.full(total + guard, 0xAA, np.uint8)
    segs, io, oo = [], 0, lead
    for f, cap in zip(frames, caps):
        segs.append((io, oo, len(f), cap)); io += len(f); oo += cap
    sa = np.array(segs, dtype=LZ4SEG_DT); res = np.zeros(len(segs), LZ4RES_DT)
    sim.sim_lz4d(cbuf.ctypes.data, obuf.ctypes.data, sa.ctypes.data, res.ctypes.data, len(segs))
    outs, oo = [], lead
    for cap in caps:
        outs.append(bytes(obuf[oo:oo + cap])); oo += cap
    return outs, res, bytes(obuf[:lead]), bytes(obuf[total:])


def test_lz4_frame_decoder_matches_the_software_path(sim):
    """K5 (round 5: a scalar walk of the token stream dealing sequences to the lanes, resolved by the batch engine of
    qzk_lz_batch.h): what LZ4F_decompress makes of the software path's frames (src/qatzip_sw.c:486-533) - every kind, the
    sizes around lz4's limits, calls above 64 KB (ONE frame of linked blocks: matches reach back across block borders),
    incompressible input (stored blocks), outputs at every 16-byte phase, several frames in one launch"""
    for kind in datagen.KINDS:
        frames, srcs = [], []
        for n in (0, 1, 12, 13, 300, 4097, 65535, 65536, 65537, 150001):
            if kind == "lzmix" and n > 70000:
                n = 66000
            src = datagen.gen_bytes(kind, n, 77)
            rc, _, fr, _ = O.sw_compress("LZ4", src, 65536, 1, cap=n + n // 255 + 4096)
            assert rc == 0
            frames.append(fr); srcs.append(src)
        for lead in (0, 5):
            outs, res, head, tail = _sim_lz4d(sim, frames, [len(s) for s in srcs], lead=lead)
            assert head == b"\xaa" * lead and tail == b"\xaa" * 64, (kind, lead)
            for i, (o, s, f) in enumerate(zip(outs, srcs, frames)):
                assert res[i]["status"] == 0 and res[i]["out_len"] == len(s) and res[i]["in_used"] == len(f), (kind, lead, i, res[i])
                assert o == s, (kind, lead, i, len(s))
    # liblz4's own frames for calls above 64 KB
    import json
    d = os.path.join(HERE, "golden", "lz4_linked")
    with open(os.path.join(d, "index.json")) as f:
        idx = json.load(f)
    frames = [open(os.path.join(d, fr["file"]), "rb").read() for fr in idx["frames"]]
    srcs = [datagen.gen_bytes(fr["kind"], fr["n"], fr["seed"]) for fr in idx["frames"]]
    outs, res, _, tail = _sim_lz4d(sim, frames, [len(s) for s in srcs])
    assert tail == b"\xaa" * 64
    for i, (o, s) in enumerate(zip(outs, srcs)):
        assert res[i]["status"] == 0 and o == s, (i, res[i])
    # shapes the batch engine has special paths for: a run (one sequence longer than a batch), short periods, long literal
    # runs between matches, matches reaching to the frame's first byte
    rng = np.random.default_rng(5)
    shapes = [b"\0" * 65536, b"ab" * 30000, b"abc" * 21000, bytes(rng.integers(0, 256, 5000, dtype=np.uint8)) * 9,
              bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) + b"x" * 5000 + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),
              b"".join(bytes([65 + (i % 7)]) * (i % 40 + 1) for i in range(3000))]
    frames = [O.sw_compress("LZ4", s, 65536, 1, cap=len(s) + 4096)[2] for s in shapes]
    outs, res, _, tail = _sim_lz4d(sim, frames, [len(s) for s in shapes])
    assert tail == b"\xaa" * 64
    for i, (o, s) in enumerate(zip(outs, shapes)):
        assert res[i]["status"] == 0 and o == s, (i, res[i])


def test_lz4_decoder_on_damaged_frames(sim):
    """flipped bits, truncation, a destination one byte short: an error (the content checksum catches what the token walk
    does not), never a byte outside the segment's output (src/qatzip_sw.c:498-500,524-532: LZ4F errors end in QZ_FAIL)"""
    import random
    errors = same = 0
    for seed in range(1, 161):
        rng = random.Random(seed)
        kind = rng.choice(datagen.KINDS)
        n = rng.choice([rng.randrange(1, 300), rng.randrange(300, 20000), rng.randrange(60000, 70000)])
        src = datagen.gen_bytes(kind, n, 1200 + seed)
        fr = bytearray(O.sw_compress("LZ4", src, 65536, 1, cap=n + n // 255 + 4096)[2])
        cap = n
        mode = rng.random()
        if mode < 0.15:
            fr = fr[:rng.randrange(1, len(fr))]
        elif mode < 0.25:
            cap = max(0, n - rng.randrange(1, 20))
        else:
            for _ in range(rng.choice([1, 1, 2, 5])):
                fr[rng.randrange(0, len(fr))] ^= 1 << rng.randrange(8)
        outs, res, head, tail = _sim_lz4d(sim, [bytes(fr)], [cap], lead=rng.choice([0, 3, 16]))
        assert tail == b"\xaa" * 64 and set(head) <= {0xAA}, seed
        if res[0]["status"] == 0:
            assert outs[0] == src[:cap] and cap == n, seed
            same += 1
        else:
            errors += 1
    assert errors > 120, (errors, same)


def test_k1_entry_cache_and_asking_ahead_variant():
    """round 6's K1 experiment stays in the kernel header behind QZK_CNBLOG / QZK_PF (profiles/r6_k1_experiments.txt): a
    wave's LDS cache of table entries, the next window's entries asked for a window ahead, exact out-of-date marks,
    deferred stores.  Built here as its own emulator library and held to the same bytes - window slides (128 KB chunks:
    the cached positions move with the window's origin), runs (every lane on one hash), table reuse across chunks."""
    so = os.path.join(SIMDIR, "libqzsim_cache.so")
    deps = [os.path.join(SIMDIR, "sim_driver.cpp"), os.path.join(ROOT, "qatzip_amd", "csrc", "qzk_deflate_lz77.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-DQZK_CNBLOG=8", "-I", SIMDIR,
                               "-Wno-unused-function", "-o", so, os.path.join(SIMDIR, "sim_driver.cpp")])
    S = C.CDLL(so)
    S.sim_deflate_fused.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
    cnt = (C.c_ulong * 2)()
    for kind, n, chunk in (("silesia", 150000, 65536), ("runs", 70000, 65536), ("text", 280000, 131072), ("lzmix", 66000, 16384)):
        src = datagen.gen_bytes(kind, n, 41)
        out, crcs = _sim_deflate(S, src, chunk, fused=True)
        assert out == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)[2], (kind, n, chunk)
        for i in range(len(crcs)):
            assert crcs[i] == (zlib.crc32(src[i * chunk:(i + 1) * chunk]) & 0xffffffff)
    S.sim_k1_counts(cnt)
    assert cnt[1] > 5000 and cnt[0] < cnt[1] // 2, list(cnt)        # most windows took the entries asked ahead, not a gather of their own
