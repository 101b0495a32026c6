#!/usr/bin/env python3
"""Generate tests/golden/manifest.json (+ small .bin fixtures).

Run ONLY in the build container: it needs the pinned third-party libraries the
reference's software path calls (system libz 1.2.11, liblz4.so.1 1.9.3) and drives
them exactly like src/qatzip_sw.c does (tests/refcalls.py).  The committed output is
data: for every case the input recipe (datagen kind/n/seed + input SHA-256), the
format / hw_buff_sz / level, and the expected compressed length + SHA-256; cases
with n <= 4096 also carry the full expected bytes in hex.

The reference's own tests hold no compressed-byte vectors; its only
known-answer check - crc out-param == zlib crc32(src) for 64 KB and 1023 B inputs
(test/main.c:4283-4337) - is recorded here as `crc32` of every input.
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402
import refcalls as R  # noqa: E402

FMTS = {"4B": R.FMT_4B, "GZIP": R.FMT_GZIP, "GZIP_EXT": R.FMT_GZIP_EXT, "RAW": R.FMT_RAW,
        "ZLIB": R.FMT_ZLIB, "LZ4": R.FMT_LZ4}


def main():
    assert R.zlib_pinned(), "need zlib 1.2.11"
    assert R.lz4_pinned(), "need liblz4 1.9.3"
    cases = []
    sizes = [0, 1, 2, 3, 4, 12, 13, 100, 1023, 4096, 16384, 65535, 65536, 65537, 131072, 200777]
    seed = 20250523
    for kind in datagen.KINDS:
        for n in sizes:
            if kind == "lzmix" and n > 65537:
                continue
            src = datagen.gen_bytes(kind, n, seed)
            for fname, fmt in FMTS.items():
                hws = [65536] if fmt == R.FMT_LZ4 else [16384, 65536, 131072]
                for hw in hws:
                    if fmt == R.FMT_LZ4 and n > 65536:
                        continue        # linked-block frames: out of round-1 scope
                    if fmt in (R.FMT_4B, R.FMT_GZIP, R.FMT_ZLIB) and hw != 65536:
                        continue
                    for level in ([1] if (fmt == R.FMT_LZ4 or kind not in ("text", "lzmix")) else [1, 2, 3]):
                        out = R.sw_compress(fmt, src, hw, level)
                        c = {"kind": kind, "n": n, "seed": seed, "fmt": fname, "hw": hw, "level": level,
                             "in_sha": datagen.sha(src), "crc32": zlib.crc32(src) & 0xffffffff,
                             "out_len": len(out), "out_sha": datagen.sha(out)}
                        if n <= 4096 and kind in ("rand", "text", "runs", "lzmix"):
                            c["out_hex"] = out.hex() if n <= 1023 else None
                        cases.append(c)
    # every zlib level the software path can be asked for (comp_lvl 1-9: greedy deflate_fast 1-3, lazy deflate_slow 4-9),
    # with the level-dependent header bytes (gzip XFL, zlib FLEVEL); 131072 crosses the 64 KB window slide
    for kind in ("text", "lzmix", "silesia", "records", "runs"):
        for n in (0, 3, 100, 4096, 65536, 65537, 200777):
            if kind == "lzmix" and n > 65537:
                continue
            src = datagen.gen_bytes(kind, n, seed + 1)
            for level in range(2, 10):
                for fname, hw in (("RAW", 65536), ("RAW", 131072), ("GZIP_EXT", 65536), ("GZIP", 65536), ("ZLIB", 65536)):
                    if (hw == 131072 and n <= 65536) or (fname in ("GZIP", "ZLIB") and n not in (0, 100, 200777)):
                        continue
                    out = R.sw_compress(FMTS[fname], src, hw, level)
                    cases.append({"kind": kind, "n": n, "seed": seed + 1, "fmt": fname, "hw": hw, "level": level,
                                  "in_sha": datagen.sha(src), "crc32": zlib.crc32(src) & 0xffffffff,
                                  "out_len": len(out), "out_sha": datagen.sha(out)})
    man = {"zlib": zlib.ZLIB_RUNTIME_VERSION, "lz4": R.lz4lib().LZ4_versionString().decode(),
           "generator": "tests/golden/gen_golden.py", "cases": cases}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(man, f, separators=(",", ":"))
    print(len(cases), "cases", os.path.getsize(os.path.join(HERE, "manifest.json")), "bytes")
    gen_lz4_linked()


def gen_lz4_linked():
    """Frames for src_len > 64 KB: LZ4F_compressFrame links the blocks of one frame (FLG 0x4C, src/qatzip_sw.c:451-456) -
    what a QZ_LZ4 session of the reference writes for such a call.  The frames themselves are committed (small), so the GPU
    box can decode them and compare its own frames with them without liblz4."""
    d = os.path.join(HERE, "lz4_linked")
    os.makedirs(d, exist_ok=True)
    idx = []
    for kind, n, seed in (("text", 65537, 11), ("text", 200777, 12), ("runs", 131072, 13), ("records", 300000, 14),
                          ("rand", 70000, 15), ("lzmix", 66000, 16), ("silesia", 262144 + 4097, 17), ("allA", 140000, 18),
                          ("mod200", 65536 * 2 + 1, 19)):
        src = datagen.gen_bytes(kind, n, seed)
        out = R.sw_compress(R.FMT_LZ4, src, 65536, 1)
        assert out[4] == 0x4c, hex(out[4])
        name = "%s_%d_%d.lz4" % (kind, n, seed)
        with open(os.path.join(d, name), "wb") as f:
            f.write(out)
        idx.append({"kind": kind, "n": n, "seed": seed, "file": name, "in_sha": datagen.sha(src), "out_len": len(out),
                    "out_sha": datagen.sha(out)})
    with open(os.path.join(d, "index.json"), "w") as f:
        json.dump({"lz4": R.lz4lib().LZ4_versionString().decode(), "frames": idx}, f, indent=1)
    print("lz4 linked frames:", len(idx), sum(i["out_len"] for i in idx), "bytes")


if __name__ == "__main__":
    main()
