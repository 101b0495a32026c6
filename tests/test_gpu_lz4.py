"""GPU parity for K4/K5 (LZ4 frames, 64 KB blocks + xxhash32): frames bit-identical to the oracle's restatement
of LZ4F_compressFrame (itself pinned to liblz4 1.9.3 goldens), decode of own and foreign frames, API path."""
import time

import pytest

import datagen
import oracle_lib as O
from qatzip_amd import api as A

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import qatzip_amd
    c = qatzip_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kind", datagen.KINDS)
def test_frames_match_oracle_and_decode(ctx, kind):
    for n, fsz in ((0, 65536), (1, 65536), (12, 65536), (13, 65536), (1000, 65536), (65536, 65536), (65535, 65536),
                   (300000, 65536), (100000, 16384), (20000, 1024)):
        if kind == "lzmix" and n > 140000:
            n = 140000
        src = datagen.gen_bytes(kind, n, 19)
        d_src = ctx.alloc(max(n, 1)); d_src.upload(src)
        nfr = max(1, (n + fsz - 1) // fsz)
        d_dst = ctx.alloc(n + nfr * 64 + 64)
        total, lens = ctx.lz4_compress_frames(d_src, n, d_dst, fsz)
        comp = d_dst.download(total).tobytes()
        exp = b"".join(O.sw_compress("LZ4", src[i * fsz:(i + 1) * fsz], 65536, 1, cap=fsz + 200)[2] for i in range(nfr))
        assert comp == exp, (kind, n, fsz)
        assert int(lens.sum()) == total
        # decode every frame in one launch
        d_out = ctx.alloc(max(n, 1))
        segs, io, oo = [], 0, 0
        for i in range(nfr):
            pl = len(src[i * fsz:(i + 1) * fsz])
            segs.append((io, oo, int(lens[i]), pl)); io += int(lens[i]); oo += pl
        res = ctx.lz4_decompress_frames(d_dst, d_out, segs)
        assert (res["status"] == 0).all() and (res["in_used"] == lens).all()
        assert d_out.download(n).tobytes() == src
        for b in (d_src, d_dst, d_out):
            b.free()


def test_decode_rejects_corruption(ctx):
    src = datagen.gen_bytes("text", 50000, 2)
    frame = O.sw_compress("LZ4", src, 65536, 1, cap=len(src) + 200)[2]
    for pos in (0, len(frame) // 2, len(frame) - 2):
        bad = bytearray(frame); bad[pos] ^= 0x41
        d_c = ctx.alloc(len(bad)); d_c.upload(bytes(bad)); d_o = ctx.alloc(len(src))
        res = ctx.lz4_decompress_frames(d_c, d_o, [(0, 0, len(bad), len(src))])
        assert res[0]["status"] != 0        # magic, payload (xxh32 mismatch) and checksum corruption are all caught
        d_c.free(); d_o.free()


def test_decode_refuses_a_block_above_the_frames_maximum(ctx):
    """LZ4F_decompress refuses a block larger than the BD byte's maximum block size; the decoder's 32-bit length sums
    rely on the same bound.  A frame that declares 64 KB blocks and carries one stored block of 70000 bytes
    is well-formed in every other respect (header checksum, end mark, content checksum)."""
    import struct
    import xxhash
    body = datagen.gen_bytes("text", 70000, 3)
    for bd, want_ok in ((0x40, False), (0x50, True)):            # 64 KB declared: refused; 256 KB declared: the same bytes decode
        hdr = bytes([0x60 | 0x04, bd])                           # version 1, independent blocks, content checksum
        frame = struct.pack("<I", 0x184D2204) + hdr + bytes([(xxhash.xxh32(hdr).intdigest() >> 8) & 0xff])
        frame += struct.pack("<I", 0x80000000 | len(body)) + body + struct.pack("<I", 0) + struct.pack("<I", xxhash.xxh32(body).intdigest())
        d_c = ctx.alloc(len(frame)); d_c.upload(frame); d_o = ctx.alloc(len(body))
        res = ctx.lz4_decompress_frames(d_c, d_o, [(0, 0, len(frame), len(body))])
        assert (res[0]["status"] == 0) == want_ok, (hex(bd), res[0])
        if want_ok:
            assert d_o.download(len(body)).tobytes() == body
        d_c.free(); d_o.free()


def test_api_lz4_session_roundtrip_and_parity():
    s = A.Session(lz4=True)
    assert s.rc_setup == A.QZ_OK
    for kind, n in (("text", 65536), ("rand", 65536), ("runs", 1000), ("silesia", 40000), ("text", 0)):
        src = datagen.gen_bytes(kind, n, 5)
        rc, used, out, _ = s.compress(src, 1, cap=n + 200)
        assert rc == A.QZ_OK and used == n
        assert out == O.sw_compress("LZ4", src, 65536, 1, cap=n + 200)[2]
        if n:
            rc, cused, back = s.decompress(out, n)
            assert rc == A.QZ_OK and back == src and cused == len(out)
    # several frames in one qzDecompress call (qzSWDecompressMultiLZ4, src/qatzip_sw.c:539-577)
    parts = [datagen.gen_bytes("silesia", 65536, 50 + i) for i in range(5)]
    comp = b"".join(s.compress(p, 1, cap=70000)[2] for p in parts)
    rc, used, back = s.decompress(comp, 5 * 65536)
    assert rc == A.QZ_OK and used == len(comp) and back == b"".join(parts)
    # above one block: ONE frame with linked blocks, as LZ4F_compressFrame writes it (src/qatzip_sw.c:451-456)
    big = datagen.gen_bytes("silesia", 300_000, 77)
    rc, used, out, _ = s.compress(big, 1, cap=len(big) + 4096)
    assert rc == A.QZ_OK and used == len(big)
    assert out == O.sw_compress("LZ4", big, 65536, 1, cap=len(big) + 4096)[2] and out[4] == 0x4c
    rc, cused, back = s.decompress(out, len(big))
    assert rc == A.QZ_OK and back == big and cused == len(out)
    # destination below LZ4F_compressFrameBound => QZ_FAIL like the software path
    rc, used, out, _ = s.compress(parts[0], 1, cap=1000)
    assert rc == A.QZ_FAIL and used == 0
    s.close()


def test_lz4_throughput_smoke(ctx):
    base = datagen.gen("silesia", 32 << 20, 3)
    d_src = ctx.alloc(base.size); d_src.upload(base)
    d_dst = ctx.alloc(base.size + 600 * 80)
    ctx.lz4_compress_frames(d_src, base.size, d_dst, 65536)
    t0 = time.perf_counter(); total, lens = ctx.lz4_compress_frames(d_src, base.size, d_dst, 65536); t1 = time.perf_counter()
    d_out = ctx.alloc(base.size)
    segs, io = [], 0
    for i in range(len(lens)):
        segs.append((io, i * 65536, int(lens[i]), 65536)); io += int(lens[i])
    ctx.lz4_decompress_frames(d_dst, d_out, segs)
    t2 = time.perf_counter(); res = ctx.lz4_decompress_frames(d_dst, d_out, segs); t3 = time.perf_counter()
    assert (res["status"] == 0).all()
    print("lz4 32 MiB: compress %.1f ms (%.2f GB/s), decompress %.1f ms (%.2f GB/s), ratio %.3f"
          % ((t1 - t0) * 1e3, base.size / (t1 - t0) / 1e9, (t3 - t2) * 1e3, base.size / (t3 - t2) / 1e9, total / base.size))
    assert total < base.size


def test_many_frames_by_persistent_waves_with_tables_in_device_memory(ctx):
    """calls of more than eight frames per CU go through qzk_lz4c_pull_kernel (persistent waves pull frame numbers, each
    wave's 16-bit hash table in device memory instead of LDS): every frame byte-identical to the one-wave-per-frame kernel
    with its table in LDS (QATZIP_AMD_LZ4_WPC=0), a sample of them to liblz4's (the oracle, src/qatzip_sw.c:443-471), ragged
    last frame, 16 KB frames too"""
    import os
    import numpy as np
    parts = [datagen.gen(k, 16 << 20, 500 + i) for i, k in enumerate(("silesia", "text", "rand", "runs", "records", "lzmix",
                                                                      "silesia", "text", "mod200", "allA"))]
    src = np.concatenate(parts)[:(150 << 20) + 4321]
    d_src = ctx.alloc(src.size); d_src.upload(src)
    for fs in (65536, 16384):
        n = src.size if fs == 65536 else (40 << 20) + 77
        nfr = (n + fs - 1) // fs
        d_a = ctx.alloc(n + nfr * 64 + 4096); d_b = ctx.alloc(n + nfr * 64 + 4096)
        old = os.environ.get("QATZIP_AMD_LZ4_WPC")
        try:
            os.environ.pop("QATZIP_AMD_LZ4_WPC", None)
            ta, la = ctx.lz4_compress_frames(d_src, n, d_a, fs)
            os.environ["QATZIP_AMD_LZ4_WPC"] = "0"
            tb, lb = ctx.lz4_compress_frames(d_src, n, d_b, fs)
        finally:
            if old is None:
                os.environ.pop("QATZIP_AMD_LZ4_WPC", None)
            else:
                os.environ["QATZIP_AMD_LZ4_WPC"] = old
        assert ta == tb and (la == lb).all()
        a = d_a.download(ta); b = d_b.download(tb)
        assert np.array_equal(a, b), fs
        offs = np.concatenate([[0], np.cumsum(la.astype(np.int64))])
        rng = np.random.default_rng(7)
        for i in list(rng.integers(0, nfr, 40)) + [0, nfr - 1]:
            piece = src[i * fs:min((i + 1) * fs, n)].tobytes()
            exp = O.sw_compress("LZ4", piece, 65536, 1, cap=len(piece) + len(piece) // 255 + 200)[2]
            assert a[offs[i]:offs[i + 1]].tobytes() == exp, (fs, i)
        d_a.free(); d_b.free()
    d_src.free()


def test_lz4_session_in_hardware_path_framing():
    """qzamd_set_hw_framing on an LZ4 session: one frame per hw_buff_sz chunk behind qzLZ4HeaderGen's header (FLG 0x4C,
    content size = consumed, src/qatzip_lz4.c:104-132) with qzLZ4FooterGen's end mark + XXH32 (:134-143); hw_buff_sz above
    64 KB gives each chunk's frame linked 64 KB blocks.  liblz4-compatible: the session's own decoder (and the oracle's)
    read it back; calls below input_sz_thrshold keep the software framing (src/qatzip.c:1934-1947)."""
    import ctypes as C
    L = A.lib()
    L.qzamd_set_hw_framing.argtypes = [C.c_void_p, C.c_int]
    for hw in (65536, 16384, 131072):
        s = A.Session(lz4=True, hw_buff_sz=hw)
        assert L.qzamd_set_hw_framing(C.byref(s.s), 1) == A.QZ_OK
        for kind, n in (("silesia", 5 * hw + 777), ("rand", 2 * hw), ("text", hw), ("allA", 3 * hw + 1)):
            src = datagen.gen_bytes(kind, n, 61)
            rc, used, out, _ = s.compress(src, 1)
            assert rc == A.QZ_OK and used == n, (hw, kind, rc)
            exp = b""
            for off in range(0, n, hw):
                piece = src[off:off + hw]
                sw = O.sw_compress("LZ4", piece, 65536, 1, cap=len(piece) + len(piece) // 255 + 4096)[2]
                desc = bytes([0x4C, 0x40]) + len(piece).to_bytes(8, "little")
                hdr = bytes([0x04, 0x22, 0x4D, 0x18]) + desc + bytes([(O.lib().qzo_xxh32(desc, len(desc), 0) >> 8) & 0xff])
                # a chunk above 64 KB is liblz4's own linked frame (its header is already this one); up to 64 KB only the
                # header differs from the software frame (0x6C: one independent block)
                assert len(piece) <= 65536 or sw[:15] == hdr
                exp += hdr + sw[15:]
            assert out == exp, (hw, kind, n, len(out), len(exp))
            rc, cused, back = s.decompress(out, n + 64)
            assert rc == A.QZ_OK and back == src and cused == len(out)
            assert O.sw_decompress("LZ4", out, n + 64)[2] == src
        small = datagen.gen_bytes("text", 1000, 3)
        assert s.compress(small, 1)[2] == O.sw_compress("LZ4", small, 65536, 1, cap=2000)[2]
        # a destination for two frames only: whole frames, QZ_BUF_ERROR with progress
        src = datagen.gen_bytes("text", 4 * hw, 5)
        full = s.compress(src, 1)[2]
        first_two = 0
        for off in (0, hw):
            piece = src[off:off + hw]
            first_two += len(O.sw_compress("LZ4", piece, 65536, 1, cap=len(piece) + 4096)[2])
        rc, used, out, _ = s.compress(src, 1, cap=first_two + 5)
        assert rc == A.QZ_BUF_ERROR and used == 2 * hw and out == full[:first_two]
        s.close()
