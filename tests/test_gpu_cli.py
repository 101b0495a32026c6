"""GPU: the qzip-style front end (qatzip_amd/cli/qzip_amd.c, counterpart of the reference's utils/qzip*.c) —
files in, files out, through qatzip.h only.  What it writes must be what the software path writes (oracle) and what
stock gzip readers accept; what stock gzip writes it must read back."""
import gzip
import os
import subprocess
import zlib

import pytest

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe():
    import qatzip_amd.build as B
    B.build()
    assert os.path.exists(B.CLI)
    return B.CLI


def run(exe, *args, stdin=None):
    return subprocess.run([exe, *args], input=stdin, capture_output=True, timeout=300)


def test_compress_file_matches_software_path_and_gzip_reads_it(exe, tmp_path):
    src = datagen.gen_bytes("silesia", 3 << 20, 11)
    f = tmp_path / "data.bin"
    f.write_bytes(src)
    os.utime(f, (1_600_000_000, 1_600_000_000))
    r = run(exe, "-k", str(f))
    assert r.returncode == 0, r.stderr
    out = (tmp_path / "data.bin.gz").read_bytes()
    assert f.exists()                                            # -k keeps the input
    assert int(os.stat(tmp_path / "data.bin.gz").st_mtime) == 1_600_000_000
    assert out == O.sw_compress("GZIP_EXT", src, 65536, 1, cap=len(src) * 9 // 8 + 65536)[2]   # one call, one member
    assert gzip.decompress(out) == src


def test_blocks_become_members_and_input_is_removed(exe, tmp_path):
    src = datagen.gen_bytes("text", 2_500_000, 3)
    f = tmp_path / "t.txt"
    f.write_bytes(src)
    r = run(exe, "-O", "gzip", "-b", "1048576", "-C", "16384", str(f))
    assert r.returncode == 0, r.stderr
    assert not f.exists()                                        # like gzip: the input goes away without -k
    out = (tmp_path / "t.txt.gz").read_bytes()
    exp = b"".join(O.sw_compress("GZIP", src[i:i + 1048576], 16384, 1)[2] for i in range(0, len(src), 1048576))
    assert out == exp
    assert gzip.decompress(out) == src                           # concatenated members
    r = run(exe, "-d", str(tmp_path / "t.txt.gz"))
    assert r.returncode == 0, r.stderr
    assert f.read_bytes() == src and not (tmp_path / "t.txt.gz").exists()


def test_gzipext_blocks_round_trip(exe, tmp_path):
    # several gzip-ext members of many chunks each in one file (what -b produces): read back in one go
    src = datagen.gen_bytes("silesia", 2_000_000, 12)
    f = tmp_path / "m.bin"
    f.write_bytes(src)
    r = run(exe, "-O", "gzipext", "-b", "524288", str(f))
    assert r.returncode == 0, r.stderr
    out = (tmp_path / "m.bin.gz").read_bytes()
    assert out == b"".join(O.sw_compress("GZIP_EXT", src[i:i + 524288], 65536, 1)[2] for i in range(0, len(src), 524288))
    r = run(exe, "-d", str(tmp_path / "m.bin.gz"))
    assert r.returncode == 0, r.stderr
    assert f.read_bytes() == src


def test_decompress_foreign_gzip(exe, tmp_path):
    src = datagen.gen_bytes("records", 1_200_000, 5)
    g = tmp_path / "foreign.gz"
    with gzip.GzipFile(filename=str(g), mode="wb", compresslevel=6) as fh:       # FNAME header, no flush markers
        fh.write(src[:700_000])
    with open(g, "ab") as fh:                                                   # second member, different producer settings
        co = zlib.compressobj(9, zlib.DEFLATED, 31)
        fh.write(co.compress(src[700_000:]) + co.flush())
    r = run(exe, "-d", "-k", str(g))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "foreign").read_bytes() == src


def test_pipe_mode_round_trip(exe):
    src = datagen.gen_bytes("lzmix", 140000, 9)
    r = run(exe, "-O", "gzipext", stdin=src)
    assert r.returncode == 0, r.stderr
    assert r.stdout == O.sw_compress("GZIP_EXT", src, 65536, 1)[2]
    r2 = run(exe, "-d", stdin=r.stdout)
    assert r2.returncode == 0 and r2.stdout == src
    r3 = run(exe, stdin=b"")                                     # empty input: the 34-byte empty member
    assert r3.returncode == 0 and len(r3.stdout) == 34 and gzip.decompress(r3.stdout) == b""


def test_level_option(exe, tmp_path):
    src = datagen.gen_bytes("text", 400_000, 4)
    for lvl in (3, 6, 9):
        f = tmp_path / ("l%d.txt" % lvl)
        f.write_bytes(src)
        r = run(exe, "-k", "-L", str(lvl), str(f))
        assert r.returncode == 0, r.stderr
        out = (tmp_path / ("l%d.txt.gz" % lvl)).read_bytes()
        assert out == O.sw_compress("GZIP_EXT", src, 65536, lvl, cap=len(src) * 9 // 8 + 65536)[2]     # zlib's level, its XFL byte
        assert gzip.decompress(out) == src
    r = run(exe, "-L", "12", str(tmp_path / "l3.txt"))                 # a QAT level zlib does not have
    assert r.returncode != 0


def test_lz4_files(exe, tmp_path):
    src = datagen.gen_bytes("silesia", 300_000, 2)
    f = tmp_path / "x.dat"
    f.write_bytes(src)
    r = run(exe, "-k", "-A", "lz4", "-O", "lz4", str(f))
    assert r.returncode == 0, r.stderr
    out = (tmp_path / "x.dat.lz4").read_bytes()
    exp = b"".join(O.sw_compress("LZ4", src[i:i + 65536], 65536, 1)[2] for i in range(0, len(src), 65536))
    assert out == exp                                            # one frame per 64 KB, each what LZ4F_compressFrame writes
    f.unlink()
    r = run(exe, "-d", str(tmp_path / "x.dat.lz4"))              # format picked from the suffix
    assert r.returncode == 0, r.stderr
    assert f.read_bytes() == src


def test_errors_are_loud(exe, tmp_path):
    bad = tmp_path / "bad.gz"
    good = O.sw_compress("GZIP_EXT", datagen.gen_bytes("text", 100000, 1), 65536, 1)[2]
    bad.write_bytes(good[:len(good) // 2])                       # truncated member
    r = run(exe, "-d", "-k", str(bad))
    assert r.returncode != 0 and not (tmp_path / "bad").exists()
    r = run(exe, "-O", "7z", str(bad))
    assert r.returncode != 0
    r = run(exe, "-d", str(tmp_path / "nosuffix"))
    assert r.returncode != 0
