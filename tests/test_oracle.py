"""Pin the CPU oracle: every committed golden vector (made by the system libz 1.2.11 /
liblz4 1.9.3 driven like src/qatzip_sw.c, tests/golden/gen_golden.py), the reference's
CRC known-answer (test/main.c:4283-4337), and round trips."""
import json
import os
import zlib

import pytest

import datagen
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "manifest.json")) as f:
    MAN = json.load(f)
_cache = {}


def _src(c):
    k = (c["kind"], c["n"], c["seed"])
    if k not in _cache:
        _cache[k] = datagen.gen_bytes(*k)
        assert datagen.sha(_cache[k]) == c["in_sha"], "datagen drifted from the golden inputs"
    return _cache[k]


def test_manifest_versions():
    assert MAN["zlib"] == "1.2.11" and MAN["lz4"] == "1.9.3"
    assert len(MAN["cases"]) > 1000


@pytest.mark.parametrize("fmt", ["RAW", "GZIP_EXT", "GZIP", "4B", "ZLIB", "LZ4"])
def test_oracle_matches_goldens(fmt):
    n_checked = 0
    for c in MAN["cases"]:
        if c["fmt"] != fmt:
            continue
        src = _src(c)
        rc, used, out, crc = O.sw_compress(fmt, src, c["hw"], c["level"], cap=len(src) * 9 // 8 + 4096)
        assert rc == 0 and used == len(src), c
        assert len(out) == c["out_len"] and datagen.sha(out) == c["out_sha"], c
        if c.get("out_hex"):
            assert out.hex() == c["out_hex"]
        n_checked += 1
    assert n_checked > 50


def test_crc_known_answer_single_chunk():
    # reference test: qzCompressCrc out == zlib crc32(src) for 64 KB and 1023 B (test/main.c:4283-4337)
    for n in (65536, 1023):
        src = datagen.gen_bytes("runs", n, 3)
        for fmt in ("GZIP_EXT", "GZIP", "RAW"):
            rc, _, _, crc = O.sw_compress(fmt, src, 65536, 1)
            assert rc == 0 and crc == (zlib.crc32(src) & 0xffffffff)
    for c in MAN["cases"][:200]:
        assert O.lib().qzo_crc32(0, _src(c), c["n"]) == c["crc32"]


def test_crc_multichunk_quirk_is_restated():
    # src/qatzip_sw.c:217-230 folds a CUMULATIVE crc with crc32_combine
    src = datagen.gen_bytes("text", 200000, 5)
    rc, _, _, crc = O.sw_compress("GZIP_EXT", src, 65536, 1)
    L = O.lib()
    exp = 0
    pos = 0
    while pos < len(src):
        pos = min(pos + 65536, len(src))
        cum = zlib.crc32(src[:pos]) & 0xffffffff
        exp = cum if exp == 0 else L.qzo_crc32_combine(exp, cum, pos)
    assert crc == exp and crc != (zlib.crc32(src) & 0xffffffff)


@pytest.mark.parametrize("fmt", ["RAW", "GZIP_EXT", "GZIP", "4B", "ZLIB", "LZ4"])
def test_oracle_roundtrip_and_system_zlib_decodes(fmt):
    for kind in datagen.KINDS:
        for n in (1, 100, 5000, 65536, 70001 if fmt != "LZ4" else 60000):
            src = datagen.gen_bytes(kind, n, 11)
            rc, used, comp, _ = O.sw_compress(fmt, src, 16384, 1, cap=n * 9 // 8 + 4096)
            assert rc == 0
            rc, cused, out = O.sw_decompress(fmt, comp, n + 16)
            assert rc == 0 and out == src and cused == len(comp)
            if fmt == "GZIP_EXT" or fmt == "GZIP":
                assert zlib.decompress(comp, 31) == src
            elif fmt == "RAW":
                assert zlib.decompress(comp, -15) == src
            elif fmt == "ZLIB":
                assert zlib.decompress(comp, 15) == src


def test_oracle_error_paths():
    src = datagen.gen_bytes("text", 3000, 1)
    rc, _, comp, _ = O.sw_compress("GZIP_EXT", src)
    bad = bytearray(comp); bad[0] ^= 0xff
    assert O.sw_decompress("GZIP_EXT", bytes(bad), 4000)[0] == -4      # QZ_DATA_ERROR
    bad = bytearray(comp); bad[len(comp) // 2] ^= 0x55
    assert O.sw_decompress("GZIP_EXT", bytes(bad), 4000)[0] != 0
    rc, _, comp, _ = O.sw_compress("LZ4", src)
    bad = bytearray(comp); bad[0] ^= 0xff
    assert O.sw_decompress("LZ4", bytes(bad), 4000)[0] == -2           # QZ_FAIL
    # compress into a too-small destination => QZ_FAIL (SW path behaviour)
    assert O.sw_compress("GZIP_EXT", src, cap=100)[0] == -2


def test_empty_input_formats():
    # src_len=0,last=1 => 34 B gzip-ext member; LZ4 => 15 B frame
    rc, used, out, _ = O.sw_compress("GZIP_EXT", b"")
    assert rc == 0 and len(out) == 34 and out[24:26] == b"\x03\x00"
    rc, used, out, _ = O.sw_compress("LZ4", b"", cap=64)
    assert rc == 0 and len(out) == 15 and out[4] == 0x64


@pytest.mark.skipif(zlib.ZLIB_RUNTIME_VERSION != "1.2.11", reason="live fuzz needs the pinned libz")
def test_live_fuzz_against_system_zlib():
    import refcalls as R
    for seed in range(12):
        for kind in ("lzmix", "text", "rand", "runs"):
            n = [777, 20000, 65536, 100000][seed % 4]
            if kind == "lzmix":
                n = min(n, 40000)
            src = datagen.gen_bytes(kind, n, 100 + seed)
            for lvl in range(1, 10):                      # 1-3 deflate_fast, 4-9 deflate_slow
                assert O.sw_compress("RAW", src, 65536, lvl)[2] == R.sw_compress(R.FMT_RAW, src, 65536, lvl), (kind, n, lvl)
    # chunk sizes either side of the 64 KB window (the slide at strstart >= 65274), last = 0, level-dependent header bytes
    for kind, n, hw in (("silesia", 300000, 131072), ("records", 600000, 524288), ("text", 70000, 16384), ("runs", 140000, 131072)):
        src = datagen.gen_bytes(kind, n, 5)
        for lvl in range(2, 10):
            assert O.sw_compress("RAW", src, hw, lvl, last=0)[2] == R.sw_compress(R.FMT_RAW, src, hw, lvl, last=0), (kind, hw, lvl)
            for fmt, rf in (("GZIP", R.FMT_GZIP), ("GZIP_EXT", R.FMT_GZIP_EXT), ("ZLIB", R.FMT_ZLIB)):
                assert O.sw_compress(fmt, src, hw, lvl)[2] == R.sw_compress(rf, src, hw, lvl), (kind, hw, lvl, fmt)
            assert R.sw_compress(R.FMT_GZIP, src, hw, lvl) == R.gzip_stream_check(src, hw, lvl)


def test_gzip_ext_members_written_by_libz_itself():
    """the pin, tightened (round 4): the goldens' GZIP_EXT members go through CPython's zlib module, which cannot pass deflate()
    a gzip header, so tests/refcalls.py typed the 24 header bytes out.  Here the system libz 1.2.11 writes the whole member
    through its C API exactly as qzDeflateSWCompress drives it (deflateSetHeader with the 'Q','Z' extra field, os 255,
    src/qatzip_sw.c:61-75,158-166) - and it is, byte for byte, the typed-out form and the oracle's."""
    import refcalls as R
    if not R.libz_pinned():
        pytest.skip("needs the system libz at the pinned version (build container)")
    n_cases = 0
    for kind in datagen.KINDS:
        for n, hw, level, last in ((0, 65536, 1, 1), (1, 65536, 1, 1), (70000, 65536, 1, 1), (200000, 65536, 1, 1), (200000, 16384, 1, 0),
                                   (300001, 131072, 6, 1), (65536, 65536, 9, 1), (131072, 65536, 3, 0)):
            if kind == "lzmix" and n > 70000:
                continue
            src = datagen.gen_bytes(kind, n, 31 + n_cases)
            got = R.libz_gzip_ext(src, hw, level, last)
            assert got == R.sw_compress(R.FMT_GZIP_EXT, src, hw, level, last), (kind, n, hw, level, last)
            rc, used, out, _ = O.sw_compress("GZIP_EXT", src, hw, level, last=last)
            assert rc == 0 and used == n and out == got, (kind, n, hw, level, last)
            n_cases += 1
    assert n_cases > 50
