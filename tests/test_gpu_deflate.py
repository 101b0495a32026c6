"""GPU parity: the HIP deflate path (K1 LZ77 + K2 Huffman, via the device C ABI) against the
CPU oracle on the same seeded inputs — bit-exact.  Needs a real MI355X."""
import zlib

import numpy as np
import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import qatzip_amd
    c = qatzip_amd.Context(0)
    yield c
    c.close()


def _gpu_raw(ctx, src, chunk, last=1, level=1):
    import qatzip_amd
    d_src = ctx.alloc(len(src)); d_src.upload(src)
    d_dst = ctx.alloc(qatzip_amd.max_deflate_len(len(src), chunk))
    n, crcs = ctx.deflate_raw(d_src, len(src), chunk, level, last, d_dst)
    out = d_dst.download(n).tobytes()
    d_src.free(); d_dst.free()
    return out, crcs


@pytest.fixture(params=["pull", "auto"])
def parse_kernel(request):
    """level 1 has two parse kernels: K1 (a wave per chunk, qzk_lz77_pull_kernel) for calls that fill the chip and K1w (a
    workgroup per chunk, qzk_lz77_wide_kernel) for launches of at most one chunk per CU.  "auto" is the product's choice
    (K1w for the small cases here); "pull" forces K1 so that both are compared with the oracle on every case."""
    import os
    old = os.environ.get("QATZIP_AMD_K1")
    if request.param == "pull":
        os.environ["QATZIP_AMD_K1"] = "pull"
    else:
        os.environ.pop("QATZIP_AMD_K1", None)
    yield request.param
    if old is None:
        os.environ.pop("QATZIP_AMD_K1", None)
    else:
        os.environ["QATZIP_AMD_K1"] = old


@pytest.mark.parametrize("kind", datagen.KINDS)
def test_deflate_raw_matches_oracle(ctx, kind, parse_kernel):
    for n, chunk in ((0, 65536), (1, 65536), (2, 65536), (3, 65536), (100, 65536), (1023, 65536), (65535, 65536),
                     (65536, 65536), (65537, 65536), (65274, 65536), (65400, 65536), (200777, 65536),
                     (70000, 16384), (300000, 131072), (1 << 20, 65536), (20000, 1024), (600000, 524288)):
        if kind == "lzmix" and n > 140000:
            n = 140000
        src = datagen.gen_bytes(kind, n, 77)
        got, crcs = _gpu_raw(ctx, src, chunk)
        rc, _, exp, _ = O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)
        assert rc == 0
        assert got == exp, (kind, n, chunk, len(got), len(exp))
        for i in range(len(crcs)):
            assert crcs[i] == (zlib.crc32(src[i * chunk:(i + 1) * chunk]) & 0xffffffff)


def test_deflate_last0_ends_with_flush_marker(ctx, parse_kernel):
    src = datagen.gen_bytes("text", 150000, 5)
    got, _ = _gpu_raw(ctx, src, 65536, last=0)
    rc, _, exp, _ = O.sw_compress("RAW", src, 65536, 1, last=0)
    assert got == exp and got[-4:] == b"\x00\x00\xff\xff"


def test_deflate_large_roundtrip_property(ctx):
    # size-independent property at scale: zlib inflates the stream back to the input (64 MiB, 1024 chunks)
    base = datagen.gen("silesia", 16 << 20, 9)
    src = np.concatenate([base, base[::-1], np.roll(base, 12345), base ^ 1]).tobytes()
    got, crcs = _gpu_raw(ctx, src, 65536)
    assert zlib.decompress(got, -15) == src
    # and a sample of chunks bit-exact against the oracle (fresh state per chunk => chunk streams are separable)
    rc, _, exp, _ = O.sw_compress("RAW", src[:4 << 20], 65536, 1, last=0)
    assert got[:len(exp)] == exp


def test_dst_too_small_is_reported(ctx):
    import qatzip_amd
    src = datagen.gen_bytes("rand", 200000, 1)
    d_src = ctx.alloc(len(src)); d_src.upload(src)
    d_dst = ctx.alloc(1000)
    with pytest.raises(qatzip_amd.QzdError):
        ctx.deflate_raw(d_src, len(src), 65536, 1, 1, d_dst)


def test_lane_per_chunk_kernel_is_bit_exact_too(ctx, monkeypatch):
    # K1b (one chunk per lane, tables in HBM): opt-in through QATZIP_AMD_DEFLATE=lane, same bytes
    monkeypatch.setenv("QATZIP_AMD_DEFLATE", "lane")
    for kind, n, chunk in (("silesia", 3 << 20, 65536), ("lzmix", 140000, 65536), ("text", 300000, 131072),
                           ("rand", 200000, 16384), ("runs", 65400, 65536)):
        src = datagen.gen_bytes(kind, n, 123)
        got, crcs = _gpu_raw(ctx, src, chunk)
        assert got == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)[2], (kind, n, chunk)


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6, 7, 8, 9])
def test_every_zlib_level_is_bit_exact(ctx, level):
    """comp_lvl 2-9 (greedy 2-3, lazy 4-9): what the software path's deflateInit2(level, ...) writes, bit for bit —
    short and empty inputs, the 65274 window slide (chunks > 64 KB), last = 0, many chunks in one call."""
    for kind, n, chunk in (("silesia", 1 << 20, 65536), ("text", 300000, 131072), ("lzmix", 100000, 65536), ("rand", 70000, 16384),
                           ("runs", 65400, 65536), ("records", 600000, 524288), ("allA", 200000, 65536), ("mod200", 9000, 1024),
                           ("text", 0, 65536), ("text", 2, 65536), ("text", 3, 65536)):
        src = datagen.gen_bytes(kind, n, 200 + level)
        got, crcs = _gpu_raw(ctx, src, chunk, level=level)
        assert got == O.sw_compress("RAW", src, chunk, level, cap=n * 9 // 8 + 65536)[2], (kind, n, chunk, level)
        if n:
            assert zlib.decompress(got, -15) == src
    src = datagen.gen_bytes("text", 150000, level)
    got, _ = _gpu_raw(ctx, src, 65536, last=0, level=level)
    assert got == O.sw_compress("RAW", src, 65536, level, last=0, cap=200000)[2] and got.endswith(b"\x00\x00\xff\xff")


@pytest.mark.parametrize("mix", ["1", "3", "5", "3072"])
def test_k1_residency_variants_are_bit_exact(monkeypatch, mix):
    # K1's persistent workgroups pull chunks from one counter and reuse their candidate table chunk after chunk;
    # force few workgroups so that every one of them does so many times
    import qatzip_amd
    monkeypatch.setenv("QATZIP_AMD_K1_WGS", mix)
    c = qatzip_amd.Context(0)
    try:
        for kind, n, chunk in (("silesia", 3 << 20, 65536), ("lzmix", 140000, 16384), ("text", 300000, 131072),
                               ("runs", 65400, 65536), ("records", 1 << 20, 65536), ("rand", 0, 65536)):
            src = datagen.gen_bytes(kind, n, 321)
            got, crcs = _gpu_raw(c, src, chunk)
            assert got == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)[2], (mix, kind, n, chunk)
    finally:
        c.close()


def test_separate_k2_and_crc_launches_are_bit_exact_too():
    """QATZIP_AMD_FUSE=0: round 1's pipeline (K2 and the chunk CRCs as launches of their own beside the next batch's K1,
    symbols per chunk of a batch) stays available for comparison - the switch is read once per process, so a fresh one"""
    import os, subprocess, sys
    code = r'''
import sys, zlib
sys.path.insert(0, "tests")
import datagen, oracle_lib as O, qatzip_amd
c = qatzip_amd.Context(0)
for kind, n, chunk in (("silesia", 3 << 20, 65536), ("text", 300000, 16384), ("rand", 70000, 65536), ("runs", 0, 65536)):
    src = datagen.gen_bytes(kind, n, 11)
    d_src = c.alloc(max(n, 1) + 512); d_dst = c.alloc(qatzip_amd.max_deflate_len(n, chunk))
    d_src.upload(src)
    ol, crcs = c.deflate_raw(d_src, n, chunk, 1, 1, d_dst)
    got = d_dst.download(ol)
    assert bytes(got) == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)[2], (kind, n, chunk)
    for i, v in enumerate(crcs):
        assert int(v) == (zlib.crc32(src[i * chunk:(i + 1) * chunk]) & 0xffffffff)
print("ok")
'''
    env = dict(os.environ, QATZIP_AMD_FUSE="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_workgroup_per_chunk_kernel_on_a_call_that_fills_the_chip():
    """QATZIP_AMD_K1=wide: K1w (qzk_deflate_wide.h: one chunk per 1024-thread workgroup, zlib's own prev[] and the chunk on
    chip, the parse as the fixpoint of assume-inserted / match / parse rounds) also for calls of more chunks than CUs
    (its persistent workgroups pull chunk after chunk) - same bytes, in a fresh process"""
    import os, subprocess, sys
    code = r'''
import sys, zlib
sys.path.insert(0, "tests")
import datagen, oracle_lib as O, qatzip_amd
c = qatzip_amd.Context(0)
for kind, n, chunk in (("silesia", 40 << 20, 65536), ("text", 300000, 16384), ("rand", 70000, 65536), ("runs", 200000, 65536), ("records", 65536 * 3 + 5, 65536), ("text", 0, 65536)):
    src = datagen.gen_bytes(kind, n, 13)
    d_src = c.alloc(max(n, 1) + 512); d_dst = c.alloc(qatzip_amd.max_deflate_len(n, chunk))
    d_src.upload(src)
    ol, crcs = c.deflate_raw(d_src, n, chunk, 1, 1, d_dst)
    assert bytes(d_dst.download(ol)) == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)[2], (kind, n, chunk)
print("ok")
'''
    env = dict(os.environ, QATZIP_AMD_K1="wide")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_stream_moved_by_the_parse_waves_themselves(ctx):
    """QATZIP_AMD_K1_OUT=launch: the waves of the one launch also put every chunk's stream in its place (front word,
    published lengths, offsets by wave prefix sums - qzk_out_advance / qzk_out_drain) instead of a scan and a gather kernel
    behind it; the product does so for calls fed from host memory.  Same bytes as the software path, chunks of very
    different cost next to each other so that the front stalls behind slow ones, a destination that is too small"""
    import os
    import qatzip_amd
    old = os.environ.get("QATZIP_AMD_K1_OUT"), os.environ.get("QATZIP_AMD_K1")
    os.environ["QATZIP_AMD_K1_OUT"] = "launch"; os.environ["QATZIP_AMD_K1"] = "pull"
    try:
        parts = []
        for i in range(400):                                   # 400 pieces of 5 kinds, 40 KiB - 104 KiB each: ~28 MiB
            kind = ("rand", "allA", "silesia", "runs", "text")[i % 5]
            parts.append(datagen.gen_bytes(kind, 40960 + 163 * i, 1000 + i))
        src = b"".join(parts)
        for chunk in (65536, 16384):
            got, crcs = _gpu_raw(ctx, src, chunk)
            exp = O.sw_compress("RAW", src, chunk, 1, cap=len(src) * 9 // 8 + (1 << 20))[2]
            assert got == exp, (chunk, len(got), len(exp))
        for n in (0, 1, 65536, 65537, 5 * 65536 + 1):
            got, _ = _gpu_raw(ctx, src[:n], 65536, last=0)
            assert got == O.sw_compress("RAW", src[:n], 65536, 1, last=0, cap=n * 9 // 8 + 65536)[2], n
        d_src = ctx.alloc(len(src)); d_src.upload(src)
        d_small = ctx.alloc(len(exp) // 2)
        with pytest.raises(qatzip_amd.QzdError):
            ctx.deflate_raw(d_src, len(src), 16384, 1, 1, d_small)
        d_src.free(); d_small.free()
    finally:
        for k, v in zip(("QATZIP_AMD_K1_OUT", "QATZIP_AMD_K1"), old):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
