"""Parity at the size that is benchmarked (BASELINE config 2: a 2 GiB call = 32768 chunks of 64 KB in ONE launch): the
reference harness' `-v` compare (test/main.c:2281-2288) at bench scale.  The oracle cannot redo 2 GiB in a test's time,
so it redoes pseudo-randomly chosen chunks: the chunk streams of a call are independent (every chunk ends byte-aligned
with a full flush), cut apart with the per-chunk lengths the device reports, each compared byte for byte with the
oracle's stream for that chunk - first and last chunk always among them - plus the stream CRC-32 against the chunks'
own.  The decode side: the whole member back through qzd_inflate_stream, CRC-32 and sampled chunks against the input.
And the 32-bit chunk epoch of the candidate table wrapping around in the middle of a call (QATZIP_AMD_EPOCH0)."""
import os
import zlib

import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
CHUNK = 65536


def _bench_buffer(ctx, total, seed=20250523):
    """the bench's buffer: 128 MiB of distinct Silesia-like data tiled with a period that is no multiple of the chunk size"""
    base = datagen.gen("silesia", min(128 << 20, total), seed)
    d_src = ctx.alloc(total)
    tile = len(base) - 4099 if total > len(base) else len(base)
    for off in range(0, total, tile):
        d_src.upload(base[:min(tile, total - off)], off)
    return d_src


def _check_sampled_chunks(ctx, d_src, d_dst, total, n_out, crcs, nsample, seed, CHUNK=CHUNK):
    nch = (total + CHUNK - 1) // CHUNK
    lens = np.zeros(nch, np.uint32)
    ctx._chk(ctx.L.qzd_chunk_lens(ctx.h, lens.ctypes.data, nch))
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    assert int(offs[-1]) == n_out
    rng = np.random.Generator(np.random.PCG64(seed))
    pick = sorted(set([0, 1, nch - 2, nch - 1]) | set(int(x) for x in rng.integers(0, nch, nsample)))
    for k in pick:
        plain = d_src.download(min(CHUNK, total - k * CHUNK), k * CHUNK).tobytes()
        got = d_dst.download(int(lens[k]), int(offs[k])).tobytes()
        exp = O.sw_compress("RAW", plain, CHUNK, 1, last=1 if k == nch - 1 else 0, cap=CHUNK * 9 // 8 + 4096)[2]
        assert got == exp, ("chunk", k, len(got), len(exp))
        assert int(crcs[k]) == (zlib.crc32(plain) & 0xffffffff), ("crc of chunk", k)
    return len(pick)


def test_one_2GiB_launch_matches_the_oracle_on_sampled_chunks():
    import qatzip_amd
    ctx = qatzip_amd.Context(0)
    total = 1 << 31
    d_src = _bench_buffer(ctx, total)
    d_dst = ctx.alloc(qatzip_amd.max_deflate_len(total, CHUNK))
    n_out, crcs = ctx.deflate_raw(d_src, total, CHUNK, 1, 1, d_dst)
    assert len(crcs) == total // CHUNK
    checked = _check_sampled_chunks(ctx, d_src, d_dst, total, n_out, crcs, 512, 1)
    # the whole stream's CRC-32 from its chunks' (crc32_combine) equals the device's own CRC kernel over the input
    crcs = np.ascontiguousarray(crcs, np.uint32)
    assert ctx.L.qzd_crc32_fold(crcs.ctypes.data, len(crcs), CHUNK, total) == ctx.crc32(d_src, total)
    # decode side at the same size: one call, 32768 segments
    d_back = ctx.alloc(total)
    iu, ol, crc = ctx.inflate_stream(d_dst, n_out, d_back, CHUNK, want_crc=True)
    assert iu == n_out and ol == total and crc == ctx.crc32(d_src, total)
    rng = np.random.Generator(np.random.PCG64(2))
    for k in [0, total // CHUNK - 1] + [int(x) for x in rng.integers(0, total // CHUNK, 256)]:
        assert np.array_equal(d_back.download(CHUNK, k * CHUNK), d_src.download(CHUNK, k * CHUNK)), ("decoded chunk", k)
    print("2 GiB call: %d chunk streams byte-identical to the oracle's, stream CRC and round trip equal" % checked)
    for b in (d_src, d_dst, d_back):
        b.free()
    ctx.close()


def test_one_4GiB_call_per_direction_is_what_the_bench_times():
    """the bench's timed region: ONE device-layer call of 4 GiB per direction (65536 chunks in one launch, 65536 segments in
    one inflate call; the device ABI's lengths are 64-bit) - sampled chunk streams against the oracle, the stream CRC-32
    from the chunks', the round trip"""
    import qatzip_amd
    ctx = qatzip_amd.Context(0)
    total = 1 << 32
    d_src = _bench_buffer(ctx, total)
    d_dst = ctx.alloc(qatzip_amd.max_deflate_len(total, CHUNK))
    n_out, crcs = ctx.deflate_raw(d_src, total, CHUNK, 1, 1, d_dst)
    assert len(crcs) == total // CHUNK
    checked = _check_sampled_chunks(ctx, d_src, d_dst, total, n_out, crcs, 256, 3)
    crcs = np.ascontiguousarray(crcs, np.uint32)
    want = ctx.crc32(d_src, total)
    assert ctx.L.qzd_crc32_fold(crcs.ctypes.data, len(crcs), CHUNK, total) == want
    d_back = ctx.alloc(total)
    iu, ol, crc = ctx.inflate_stream(d_dst, n_out, d_back, CHUNK, want_crc=True)
    assert iu == n_out and ol == total and crc == want
    rng = np.random.Generator(np.random.PCG64(4))
    for k in [0, 1, total // CHUNK - 2, total // CHUNK - 1] + [int(x) for x in rng.integers(0, total // CHUNK, 256)]:
        assert np.array_equal(d_back.download(CHUNK, k * CHUNK), d_src.download(CHUNK, k * CHUNK)), ("decoded chunk", k)
    print("4 GiB call: %d chunk streams byte-identical to the oracle's, stream CRC, round trip and 260 decoded chunks equal" % checked)
    for b in (d_src, d_dst, d_back):
        b.free()
    ctx.close()


@pytest.mark.parametrize("chunk", [16384, 131072])
def test_raw_sweep_at_call_scale(chunk):
    """BASELINE config 3 at the size the bench runs it: 1 GiB of QZ_DEFLATE_RAW at hw_buff_sz 16 KB (65536 segments: the
    decoder's default rule gives each four lanes) and 128 KB (8192 segments of sixteen lanes) - sampled chunk streams
    against the oracle, the stream's CRC from the chunks', the round trip through the default decode path (no environment
    override), sampled decoded chunks against the input"""
    import qatzip_amd
    ctx = qatzip_amd.Context(0)
    total = 1 << 30
    d_src = _bench_buffer(ctx, total)
    d_dst = ctx.alloc(qatzip_amd.max_deflate_len(total, chunk))
    n_out, crcs = ctx.deflate_raw(d_src, total, chunk, 1, 1, d_dst)
    assert len(crcs) == total // chunk
    checked = _check_sampled_chunks(ctx, d_src, d_dst, total, n_out, crcs, 192 if chunk < 65536 else 48, 5, CHUNK=chunk)
    crcs = np.ascontiguousarray(crcs, np.uint32)
    want = ctx.crc32(d_src, total)
    assert ctx.L.qzd_crc32_fold(crcs.ctypes.data, len(crcs), chunk, total) == want
    d_back = ctx.alloc(total)
    iu, ol, crc = ctx.inflate_stream(d_dst, n_out, d_back, chunk, want_crc=True)
    assert iu == n_out and ol == total and crc == want
    rng = np.random.Generator(np.random.PCG64(6))
    nch = total // chunk
    for k in [0, 1, nch - 2, nch - 1] + [int(x) for x in rng.integers(0, nch, 128)]:
        assert np.array_equal(d_back.download(chunk, k * chunk), d_src.download(chunk, k * chunk)), ("decoded chunk", k)
    print("1 GiB RAW at %d: %d chunk streams byte-identical to the oracle's, round trip equal" % (chunk, checked))
    for b in (d_src, d_dst, d_back):
        b.free()
    ctx.close()


def test_lz4_frames_at_call_scale():
    """BASELINE config 4 at the bench's size: 1 GiB as 16384 LZ4 frames of one 64 KB block with content size and XXH32 -
    sampled frames against the oracle's LZ4F_compressFrame, every frame decoded (XXH32 verified in-kernel), the output's
    CRC-32 and sampled blocks against the input"""
    import qatzip_amd
    ctx = qatzip_amd.Context(0)
    total = 1 << 30
    nfr = total // CHUNK
    d_src = _bench_buffer(ctx, total)
    d_c = ctx.alloc(total + nfr * 64 + 4096)
    cl, lens = ctx.lz4_compress_frames(d_src, total, d_c, CHUNK)
    offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    assert int(offs[-1]) == cl
    rng = np.random.Generator(np.random.PCG64(7))
    pick = sorted(set([0, 1, nfr - 2, nfr - 1]) | set(int(x) for x in rng.integers(0, nfr, 384)))
    for k in pick:
        plain = d_src.download(CHUNK, k * CHUNK).tobytes()
        got = d_c.download(int(lens[k]), int(offs[k])).tobytes()
        assert got == O.sw_compress("LZ4", plain, CHUNK, 1, cap=CHUNK + 200)[2], ("frame", k)
    segs = np.zeros(nfr, qatzip_amd._lib.LZ4SEG_DT)
    segs["in_off"] = offs[:-1]; segs["out_off"] = np.arange(nfr, dtype=np.int64) * CHUNK; segs["in_len"] = lens; segs["out_cap"] = CHUNK
    res = np.zeros(nfr, qatzip_amd._lib.LZ4RES_DT)
    d_back = ctx.alloc(total)
    ctx._chk(ctx.L.qzd_lz4_decompress_frames(ctx.h, d_c.ptr, d_back.ptr, segs.ctypes.data, nfr, res.ctypes.data))
    assert (res["status"] == 0).all() and (res["out_len"] == CHUNK).all() and (res["in_used"] == lens).all()
    assert ctx.crc32(d_back, total) == ctx.crc32(d_src, total)
    for k in pick[::3]:
        assert np.array_equal(d_back.download(CHUNK, k * CHUNK), d_src.download(CHUNK, k * CHUNK)), ("decoded block", k)
    print("1 GiB of LZ4 frames: %d frames byte-identical to the oracle's, all decoded, CRC equal" % len(pick))
    for b in (d_src, d_c, d_back):
        b.free()
    ctx.close()


def test_inflate_of_an_oracle_made_member_at_call_scale():
    """the decoder on streams it did not write: 4096 chunk streams made by the ORACLE (256 MiB, what the oracle does in
    the test's time), concatenated as the software path lays them out, decoded in one call through the two-phase path"""
    import qatzip_amd
    ctx = qatzip_amd.Context(0)
    n = 64 << 20
    src = datagen.gen_bytes("silesia", n, 99)
    rc, used, comp, _ = O.sw_compress("RAW", src, CHUNK, 1, cap=n * 9 // 8 + 65536)
    assert rc == 0 and used == n
    reps = 4                                                   # four such streams side by side: 4096 segments in the call
    whole = b"".join(O.sw_compress("RAW", src[i * (n // reps):] + src[:i * (n // reps)], CHUNK, 1, last=1 if i == reps - 1 else 0,
                                   cap=n * 9 // 8 + 65536)[2] for i in range(reps))
    plain = b"".join(src[i * (n // reps):] + src[:i * (n // reps)] for i in range(reps))
    d_c = ctx.alloc(len(whole)); d_c.upload(whole)
    d_o = ctx.alloc(len(plain))
    os.environ["QATZIP_AMD_INFLATE"] = "lane"                  # the path a 2 GiB call takes, at a size the oracle can make
    try:
        iu, ol, crc = ctx.inflate_stream(d_c, len(whole), d_o, CHUNK, want_crc=True)
    finally:
        del os.environ["QATZIP_AMD_INFLATE"]
    assert iu == len(whole) and ol == len(plain) and crc == (zlib.crc32(plain) & 0xffffffff)
    assert d_o.download(len(plain)).tobytes() == plain
    d_c.free(); d_o.free(); ctx.close()


def test_table_epoch_wraps_in_the_middle_of_a_call():
    """K1's candidate table tags every entry with its chunk's 32-bit epoch; the pool forgets everything once before the
    counter would wrap.  QATZIP_AMD_EPOCH0 starts the counter just below the wrap so that a small call crosses it."""
    import subprocess
    import sys
    code = r'''
import os, sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, datagen, oracle_lib as O, qatzip_amd
ctx = qatzip_amd.Context(0)
for rep, n in enumerate((200 * 65536 + 17, 300 * 65536, 65536)):
    src = datagen.gen_bytes("silesia", n, 300 + rep)
    d_src = ctx.alloc(n); d_src.upload(src)
    d_dst = ctx.alloc(qatzip_amd.max_deflate_len(n, 65536))
    n_out, crcs = ctx.deflate_raw(d_src, n, 65536, 1, 1, d_dst)
    exp = O.sw_compress("RAW", src, 65536, 1, cap=n * 9 // 8 + 65536)[2]
    assert d_dst.download(n_out).tobytes() == exp, (rep, n_out, len(exp))
print("ok")
''' % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, QATZIP_AMD_EPOCH0=str(0xffffffff - 250))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
