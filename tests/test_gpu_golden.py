"""Golden-direct parity on the GPU box: the HIP path through qatzip.h against tests/golden/manifest.json itself (outputs
of the real libz 1.2.11 / liblz4 1.9.3 driven like src/qatzip_sw.c:77-256,443-471, made by tests/golden/gen_golden.py) -
length and SHA-256 of every case, all formats, levels and chunk sizes.  No oracle library in this chain: the shipped
oracle .so is not what these assertions rest on.  Plus: the pin of the oracle itself (a subset of tests/test_oracle.py)
runs here too, and the frames liblz4 writes for calls above 64 KB (linked blocks, FLG 0x4C) are decoded."""
import json
import os

import pytest

import datagen
from qatzip_amd import api as A

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "manifest.json")) as f:
    MAN = json.load(f)
with open(os.path.join(HERE, "golden", "lz4_linked", "index.json")) as f:
    LZ4L = json.load(f)
_cache = {}


def _src(c):
    k = (c["kind"], c["n"], c["seed"])
    if k not in _cache:
        if len(_cache) > 64:
            _cache.clear()
        _cache[k] = datagen.gen_bytes(*k)
        assert datagen.sha(_cache[k]) == c["in_sha"], "datagen drifted from the golden inputs"
    return _cache[k]


def _session(fmt, hw, level):
    if fmt == "LZ4":
        return A.Session(hw_buff_sz=hw, comp_lvl=level, lz4=True)
    if fmt == "ZLIB":
        return A.Session(hw_buff_sz=hw, comp_lvl=level, zlib_format=True)
    code = {"4B": A.QZ_DEFLATE_4B, "GZIP": A.QZ_DEFLATE_GZIP, "GZIP_EXT": A.QZ_DEFLATE_GZIP_EXT, "RAW": A.QZ_DEFLATE_RAW}[fmt]
    return A.Session(code, hw, comp_lvl=level)


@pytest.mark.parametrize("fmt", ["RAW", "GZIP_EXT", "GZIP", "4B", "ZLIB", "LZ4"])
def test_hip_output_equals_the_goldens(fmt):
    sessions, n_checked = {}, 0
    for c in MAN["cases"]:
        if c["fmt"] != fmt:
            continue
        key = (c["hw"], c["level"])
        if key not in sessions:
            sessions[key] = _session(fmt, c["hw"], c["level"])
            assert sessions[key].rc_setup == A.QZ_OK
        src = _src(c)
        rc, used, out, crc = sessions[key].compress(src, 1, crc0=0)
        assert rc == A.QZ_OK and used == len(src), (c, rc)
        assert len(out) == c["out_len"] and datagen.sha(out) == c["out_sha"], {k: c[k] for k in ("kind", "n", "fmt", "hw", "level")}
        if c.get("out_hex"):
            assert out.hex() == c["out_hex"]
        n_checked += 1
    for s in sessions.values():
        s.close()
    assert n_checked > 50
    print("%s: %d golden cases, HIP output identical" % (fmt, n_checked))


def test_oracle_pin_holds_on_this_box():
    """the checker the other GPU tests compare with, against a slice of the same goldens (every 7th case)"""
    import oracle_lib as O
    n = 0
    for c in MAN["cases"][::7]:
        src = _src(c)
        rc, used, out, crc = O.sw_compress(c["fmt"], src, c["hw"], c["level"], cap=len(src) * 9 // 8 + 4096)
        assert rc == 0 and used == len(src) and len(out) == c["out_len"] and datagen.sha(out) == c["out_sha"], c
        n += 1
    assert n > 300


def test_calls_above_64k_are_liblz4s_linked_frames():
    """compress side: a QZ_LZ4 call above 64 KB gives the bytes liblz4 1.9.3 gives (FLG 0x4C, linked blocks)"""
    import oracle_lib as O
    s = A.Session(lz4=True)
    for fr in LZ4L["frames"]:
        with open(os.path.join(HERE, "golden", "lz4_linked", fr["file"]), "rb") as f:
            exp = f.read()
        src = datagen.gen_bytes(fr["kind"], fr["n"], fr["seed"])
        rc, used, out, _ = s.compress(src, 1)
        assert rc == A.QZ_OK and used == len(src) and out == exp, (fr["file"], rc, len(out), len(exp))
    for kind, n, seed in (("silesia", 1 << 20, 3), ("text", 65536 + 12, 4), ("records", 5 * 65536 + 13, 5), ("rand", 200000, 6)):
        src = datagen.gen_bytes(kind, n, seed)
        rc, used, out, _ = s.compress(src, 1)
        exp = O.sw_compress("LZ4", src, 65536, 1, cap=n + n // 255 + 4096)[2]
        assert rc == A.QZ_OK and out == exp and out[4] == 0x4c, (kind, n, rc)
        rc, cused, back = s.decompress(out, n + 64)
        assert rc == A.QZ_OK and back == src and cused == len(out)
    s.close()


def test_linked_block_lz4_frames_decode():
    """what LZ4F_compressFrame writes for src_len > 64 KB (src/qatzip_sw.c:451-456): one frame, linked blocks (a block
    may reach 64 KB back into the blocks before it), content size + content checksum - liblz4 1.9.3's own bytes"""
    s = A.Session(lz4=True)
    whole, srcs = b"", b""
    for fr in LZ4L["frames"]:
        with open(os.path.join(HERE, "golden", "lz4_linked", fr["file"]), "rb") as f:
            comp = f.read()
        assert comp[4] == 0x4c and len(comp) == fr["out_len"] and datagen.sha(comp) == fr["out_sha"]
        src = datagen.gen_bytes(fr["kind"], fr["n"], fr["seed"])
        assert datagen.sha(src) == fr["in_sha"]
        rc, used, out = s.decompress(comp, fr["n"] + 64)
        assert rc == A.QZ_OK and used == len(comp) and out == src, (fr["file"], rc, used, len(out))
        whole += comp; srcs += src
    rc, used, out = s.decompress(whole, len(srcs) + 64)            # all of them back to back, one call
    assert rc == A.QZ_OK and used == len(whole) and out == srcs
    bad = bytearray(whole[:200000]); bad[150000] ^= 0x40           # damage inside a later block of the second frame
    rc, used, out = s.decompress(bytes(bad), len(srcs) + 64)
    assert rc == A.QZ_FAIL or (rc == A.QZ_OK and len(out) < len(srcs) and srcs.startswith(out))   # whole frames before it, or the error
    s.close()
