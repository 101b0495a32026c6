"""GPU parity for K3 (raw inflate) + K6 (CRC-32): decode streams produced by the oracle / system zlib /
our own compressor and compare with the known plaintext - bit-exact; plus the segment-chain logic."""
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


def _inflate(ctx, comp, cap, hint=65536):
    d_src = ctx.alloc(len(comp)); d_src.upload(comp)
    d_dst = ctx.alloc(max(cap, 1))
    try:
        iu, ol, crc = ctx.inflate_stream(d_src, len(comp), d_dst, hint)
        out = d_dst.download(ol).tobytes()
    finally:
        d_src.free(); d_dst.free()
    return iu, out, crc


@pytest.mark.parametrize("kind", datagen.KINDS)
def test_inflate_oracle_streams(ctx, kind):
    for n, chunk in ((1, 65536), (1000, 65536), (65536, 65536), (200777, 65536), (70000, 16384), (300000, 131072)):
        if kind == "lzmix" and n > 140000:
            n = 140000
        src = datagen.gen_bytes(kind, n, 21)
        rc, _, comp, _ = O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)
        assert rc == 0
        for hint in (chunk, 65536, 0):                   # right hint, wrong hint (two-pass), no hint
            iu, out, crc = _inflate(ctx, comp, n, hint)
            assert out == src, (kind, n, chunk, hint)
            assert iu == len(comp) and crc == (zlib.crc32(src) & 0xffffffff)


def test_inflate_foreign_zlib_levels(ctx):
    # streams from other producers: no flush markers, long codes, stored blocks, sync flushes with history
    src = datagen.gen_bytes("silesia", 300000, 4)
    for lvl in (0, 1, 6, 9):
        co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
        comp = co.compress(src) + co.flush()
        iu, out, crc = _inflate(ctx, comp, len(src))
        assert out == src and iu == len(comp)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = b""
    for i in range(0, len(src), 50000):
        comp += co.compress(src[i:i + 50000]) + co.flush(zlib.Z_SYNC_FLUSH)     # history crosses the markers
    comp += co.flush()
    iu, out, crc = _inflate(ctx, comp, len(src))
    assert out == src and iu == len(comp)


def test_inflate_marker_pattern_inside_data(ctx):
    # plaintext full of 00 00 FF FF, stored (incompressible wrapper) so the pattern survives into the stream
    rnd = datagen.gen("rand", 40000, 3)
    rnd[1000:1004] = [0, 0, 0xff, 0xff]
    rnd[20000:20004] = [0, 0, 0xff, 0xff]
    src = rnd.tobytes() * 3
    rc, _, comp, _ = O.sw_compress("RAW", src, 65536, 1, cap=len(src) * 2)
    assert comp.count(b"\x00\x00\xff\xff") > len(src) // 65536
    iu, out, crc = _inflate(ctx, comp, len(src))
    assert out == src and iu == len(comp)


def test_inflate_errors(ctx):
    import qatzip_amd
    src = datagen.gen_bytes("text", 100000, 8)
    rc, _, comp, _ = O.sw_compress("RAW", src, 65536, 1)
    bad = bytearray(comp); bad[len(bad) // 3] ^= 0x5a
    try:
        iu, out, crc = _inflate(ctx, bytes(bad), len(src))
        assert out != src or crc != (zlib.crc32(src) & 0xffffffff)      # corruption must not pass silently
    except qatzip_amd.QzdError:
        pass
    with pytest.raises(qatzip_amd.QzdError):
        _inflate(ctx, comp, 5000)                                       # destination too small
    with pytest.raises(qatzip_amd.QzdError):
        _inflate(ctx, comp[:len(comp) // 2], len(src))                  # truncated input


def test_roundtrip_own_compressor_large(ctx):
    import qatzip_amd
    base = datagen.gen("silesia", 8 << 20, 13)
    src = np.concatenate([base, base[::-1], base ^ 3]).tobytes()          # 24 MiB, 384 chunks
    d_src = ctx.alloc(len(src)); d_src.upload(src)
    d_c = ctx.alloc(qatzip_amd.max_deflate_len(len(src)))
    n, crcs = ctx.deflate_raw(d_src, len(src), 65536, 1, 1, d_c)
    d_o = ctx.alloc(len(src))
    iu, ol, crc = ctx.inflate_stream(d_c, n, d_o, 65536)
    assert iu == n and ol == len(src) and crc == (zlib.crc32(src) & 0xffffffff)
    assert d_o.download(ol).tobytes() == src
    assert ctx.crc32(d_src, len(src)) == (zlib.crc32(src) & 0xffffffff)


@pytest.mark.parametrize("mode", ["lane", "wave"])
def test_both_inflate_kernels_agree(ctx, monkeypatch, mode):
    # K3 (segment per wave) and K3b (segment per lane) are selected by segment count; force each one
    monkeypatch.setenv("QATZIP_AMD_INFLATE", mode)
    for kind, n, chunk in (("silesia", 3 << 20, 65536), ("lzmix", 140000, 65536), ("rand", 300000, 65536),
                           ("allA", 1 << 20, 16384)):
        src = datagen.gen_bytes(kind, n, 33)
        rc, _, comp, _ = O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)
        for hint in (chunk, 0):
            iu, out, crc = _inflate(ctx, comp, n, hint)
            assert out == src and iu == len(comp) and crc == (zlib.crc32(src) & 0xffffffff), (mode, kind, hint)
    co = zlib.compressobj(9, zlib.DEFLATED, -15)
    src = datagen.gen_bytes("text", 500000, 1)
    comp = co.compress(src) + co.flush()
    iu, out, crc = _inflate(ctx, comp, len(src))
    assert out == src


@pytest.mark.parametrize("occ", ["2", "3"])
@pytest.mark.parametrize("k", ["4", "8", "16", "32"])
def test_speculative_phase_a_is_bit_exact(ctx, monkeypatch, k, occ):
    # sub-segment speculation (qzk_inflate_spec.h): K lanes per segment; whatever it cannot take goes through
    # the serial kernel, so every kind of stream must still come out right - with the kernel's register budget cut for two
    # waves a SIMD and for three (a third of the registers spilled; what launches of more than sixteen waves a CU take)
    if k == "4" and occ == "3":
        pytest.skip("four lanes a segment: the root tables' LDS allows two waves a SIMD only")
    monkeypatch.setenv("QATZIP_AMD_INFLATE", "lane")
    monkeypatch.setenv("QATZIP_AMD_INFLATE_K", k)
    monkeypatch.setenv("QATZIP_AMD_INFLATE_OCC", occ)
    for kind, n, chunk in (("silesia", 6 << 20, 65536), ("text", 3 << 20, 65536), ("lzmix", 140000, 65536),
                           ("rand", 300000, 65536), ("runs", 1 << 20, 16384), ("records", 2 << 20, 131072), ("allA", 1 << 20, 65536)):
        src = datagen.gen_bytes(kind, n, 57)
        rc, _, comp, _ = O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)
        for hint in (chunk, 0):                              # optimistic pass (hints) and the two-pass chain (exact lengths)
            iu, out, crc = _inflate(ctx, comp, n, hint)
            assert out == src and iu == len(comp) and crc == (zlib.crc32(src) & 0xffffffff), (k, occ, kind, hint)
    src = datagen.gen_bytes("text", 500000, 2)               # a foreign stream: one long member, no markers
    co = zlib.compressobj(9, zlib.DEFLATED, -15)
    comp = co.compress(src) + co.flush()
    iu, out, crc = _inflate(ctx, comp, len(src))
    assert out == src


def test_two_sessions_on_one_gpu_at_once():
    """Two contexts driven by two host threads at the same time (the reference's harness shape, test/main.c:2175-2202;
    bench.py's concurrent_sessions leg): the compress side shares the device's tables and scratch under a lock, the decode
    side has per-context scratch - both threads must get their own bytes back."""
    import threading
    import qatzip_amd
    errs = []

    def body(seed):
        try:
            c = qatzip_amd.Context(0)
            for rep in range(3):
                n, chunk = (5 << 20) + 12345 * seed + rep, 65536
                src = datagen.gen_bytes("silesia" if seed & 1 else "text", n, 100 + seed + rep)
                d_src = c.alloc(n); d_src.upload(src)
                d_comp = c.alloc(qatzip_amd.max_deflate_len(n, chunk)); d_back = c.alloc(n)
                cl, _ = c.deflate_raw(d_src, n, chunk, 1, 1, d_comp)
                assert d_comp.download(cl).tobytes() == O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536)[2]
                iu, ol, crc = c.inflate_stream(d_comp, cl, d_back, chunk, want_crc=True)
                assert iu == cl and ol == n and d_back.download(n).tobytes() == src
                d_src.free(); d_comp.free(); d_back.free()
            c.close()
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=body, args=(s,)) for s in (1, 2, 3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_decode_without_room_for_its_scratch(ctx, monkeypatch):
    """the K-lane phase A wants ~10 bytes of token scratch per output byte.  When that cannot be had the decode
    goes on with one sub-stream per segment (a third of it), then with a wave per segment (none) - the caller gets its bytes,
    not QZ_DATA_ERROR for valid data.  QATZIP_AMD_SCRATCH_MAX stands in for the allocation that fails."""
    n = 8 << 20
    src = datagen.gen_bytes("silesia", n, 31)
    rc, _, comp, _ = O.sw_compress("RAW", src, 65536, 1, cap=n * 9 // 8 + 65536)
    assert rc == 0
    for cap_bytes in (48 << 20, 1 << 20):          # room for one sub-stream per segment (K = 1: ~30 MB); room for nothing
        monkeypatch.setenv("QATZIP_AMD_SCRATCH_MAX", str(cap_bytes))
        iu, out, crc = _inflate(ctx, comp, n, 65536)
        assert out == src and iu == len(comp) and crc == (zlib.crc32(src) & 0xffffffff), cap_bytes
    monkeypatch.delenv("QATZIP_AMD_SCRATCH_MAX")
    iu, out, crc = _inflate(ctx, comp, n, 65536)
    assert out == src


def test_marker_scan_finds_what_the_reference_scan_finds(ctx, monkeypatch):
    """the 00 00 FF FF scan (sixteen positions a lane through two 8-byte loads and a cross-lane read) against the scan it
    replaced, over streams with markers at every alignment, at the buffer's very end, in the wave's last lane, and with
    false ones inside the data: QATZIP_AMD_MARKER_CHECK makes the device layer run both and fail on any difference - a
    missed marker would only send the decode down a slower path, which no parity test sees"""
    monkeypatch.setenv("QATZIP_AMD_MARKER_CHECK", "1")
    rng = np.random.default_rng(9)
    for chunk, n in ((1000, 300000), (1021, 290000), (4096, 1 << 20), (65536, 3 << 20), (333, 100001)):
        src = datagen.gen_bytes("silesia", n, 77)
        rc, _, comp, _ = O.sw_compress("RAW", src, chunk, 1, cap=n * 9 // 8 + 65536 + 8 * (n // chunk + 1))
        assert rc == 0
        iu, out, crc = _inflate(ctx, comp, n, chunk)
        assert out == src and iu == len(comp) and crc == (zlib.crc32(src) & 0xffffffff), (chunk, n)
    # stored blocks full of false markers (00 00 FF FF inside the data), at every byte phase
    raw = bytearray(rng.integers(0, 256, 200000, dtype=np.uint8))
    for i in range(0, len(raw) - 8, 37):
        raw[i:i + 4] = b"\x00\x00\xff\xff"
    raw[-4:] = b"\x00\x00\xff\xff"
    src = bytes(raw)
    rc, _, comp, _ = O.sw_compress("RAW", src, 65536, 1, cap=len(src) * 9 // 8 + 65536)
    assert rc == 0
    iu, out, crc = _inflate(ctx, comp, len(src), 65536)
    assert out == src and iu == len(comp)
