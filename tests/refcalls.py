"""Drive the SYSTEM libz / liblz4 exactly the way the reference's software path does.

Only usable in the build container (needs libz 1.2.11 + liblz4.so.1 1.9.3); used by
tests/golden/gen_golden.py to create the committed fixtures and, when the libraries
are present at the pinned versions, by tests/test_oracle.py for extra fuzzing.

Call sites restated (reference = intel/QATzip):
  qzDeflateSWCompress  src/qatzip_sw.c:131-253  deflateInit2(lvl, Z_DEFLATED, wbits, 9, 0),
                       deflate(Z_FULL_FLUSH) per hw_buff_sz chunk, deflate(Z_FINISH) on the last
  qzLZ4SWCompress      src/qatzip_sw.c:451-456  LZ4F_compressFrame(prefs{contentChecksum=1,
                       contentSize=src_len, autoFlush=1, compressionLevel=lvl})
"""
import ctypes
import struct
import zlib

FMT_4B, FMT_GZIP, FMT_GZIP_EXT, FMT_RAW, FMT_LZ4, FMT_LZ4S, FMT_ZLIB = range(7)


def zlib_pinned():
    return zlib.ZLIB_RUNTIME_VERSION == "1.2.11"


def _piece(co, chunk: bytes, fin: bool) -> bytes:
    """deflate(chunk) + Z_FULL_FLUSH / Z_FINISH the way the reference issues it (one deflate() call into a large
    destination, src/qatzip_sw.c:190-197)."""
    piece = co.compress(chunk) + co.flush(zlib.Z_FINISH if fin else zlib.Z_FULL_FLUSH)
    # CPython's flush() calls deflate() again whenever its own output buffer came back exactly full, and zlib answers a
    # repeated Z_FULL_FLUSH with a second empty stored block (zlib.h: "avail_out greater than six to avoid repeated flush
    # markers").  The reference never sees that; drop the artefact.
    if not fin and len(chunk) and piece.endswith(b"\x00\x00\xff\xff\x00\x00\x00\xff\xff"):
        piece = piece[:-5]
    return piece


def raw_chunks(src: bytes, hw: int, level: int = 1, last: int = 1):
    """Per-chunk raw deflate pieces of ONE continuous stream (wbits -15)."""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY)
    out = []
    n = len(src)
    pos = 0
    while True:
        send = min(hw, n - pos)
        chunk = src[pos:pos + send]
        pos += send
        fin = (pos == n and last == 1)
        piece = _piece(co, chunk, fin)
        out.append(piece)
        if pos == n:
            break
    return out


def sw_compress(fmt: int, src: bytes, hw: int = 65536, level: int = 1, last: int = 1) -> bytes:
    if fmt == FMT_LZ4:
        return lz4f_compress_frame(src, level)
    raw = b"".join(raw_chunks(src, hw, level, last))
    xfl = 2 if level == 9 else (4 if level < 2 else 0)
    crc = zlib.crc32(src) & 0xffffffff
    if fmt == FMT_RAW:
        return raw
    if fmt == FMT_4B:
        hdr = struct.pack("<I", len(raw)) if last else b"\0\0\0\0"
        return hdr + raw
    if fmt == FMT_GZIP:
        out = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, xfl, 3]) + raw
        return out + (struct.pack("<II", crc, len(src) & 0xffffffff) if last else b"")
    if fmt == FMT_GZIP_EXT:
        sizes = struct.pack("<II", len(src), len(raw)) if last else b"\0" * 8
        out = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, xfl, 255, 12, 0]) + b"QZ\x08\x00" + sizes + raw
        return out + (struct.pack("<II", crc, len(src) & 0xffffffff) if last else b"")
    if fmt == FMT_ZLIB:
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_DEFAULT_STRATEGY)
        out = b""
        pos = 0
        while True:
            send = min(hw, len(src) - pos)
            chunk = src[pos:pos + send]
            pos += send
            fin = (pos == len(src) and last == 1)
            out += _piece(co, chunk, fin)
            if pos == len(src):
                return out
    raise ValueError(fmt)


def gzip_stream_check(src: bytes, hw: int, level: int = 1) -> bytes:
    """The same bytes produced by zlib's own gzip wrapper (wbits 31) - cross-check of sw_compress(FMT_GZIP)."""
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 9, zlib.Z_DEFAULT_STRATEGY)
    out = b""
    pos = 0
    while True:
        send = min(hw, len(src) - pos)
        chunk = src[pos:pos + send]
        pos += send
        fin = pos == len(src)
        out += _piece(co, chunk, fin)
        if fin:
            return out


# ---------------------------------------------------------------- lz4 via ctypes
_lz4 = None


class _FrameInfo(ctypes.Structure):
    _fields_ = [("blockSizeID", ctypes.c_int), ("blockMode", ctypes.c_int),
                ("contentChecksumFlag", ctypes.c_int), ("frameType", ctypes.c_int),
                ("contentSize", ctypes.c_ulonglong), ("dictID", ctypes.c_uint),
                ("blockChecksumFlag", ctypes.c_int)]


class _Prefs(ctypes.Structure):
    _fields_ = [("frameInfo", _FrameInfo), ("compressionLevel", ctypes.c_int),
                ("autoFlush", ctypes.c_uint), ("favorDecSpeed", ctypes.c_uint),
                ("reserved", ctypes.c_uint * 3)]


def lz4lib():
    global _lz4
    if _lz4 is None:
        try:
            lib = ctypes.CDLL("liblz4.so.1")
        except OSError:
            _lz4 = False
            return None
        lib.LZ4_versionString.restype = ctypes.c_char_p
        lib.LZ4F_compressFrameBound.restype = ctypes.c_size_t
        lib.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
        lib.LZ4F_compressFrame.restype = ctypes.c_size_t
        lib.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p]
        lib.LZ4F_isError.restype = ctypes.c_uint
        lib.LZ4F_isError.argtypes = [ctypes.c_size_t]
        lib.LZ4_compress_default.restype = ctypes.c_int
        lib.LZ4_compress_default.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.LZ4_decompress_safe.restype = ctypes.c_int
        lib.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _lz4 = lib
    return _lz4 or None


def lz4_pinned():
    lib = lz4lib()
    return bool(lib) and lib.LZ4_versionString() == b"1.9.3"


def lz4f_compress_frame(src: bytes, level: int = 1) -> bytes:
    lib = lz4lib()
    prefs = _Prefs()
    prefs.frameInfo.contentChecksumFlag = 1
    prefs.frameInfo.contentSize = len(src)
    prefs.autoFlush = 1
    prefs.compressionLevel = level
    cap = lib.LZ4F_compressFrameBound(len(src), ctypes.byref(prefs))
    dst = ctypes.create_string_buffer(cap)
    r = lib.LZ4F_compressFrame(dst, cap, src, len(src), ctypes.byref(prefs))
    if lib.LZ4F_isError(r):
        raise RuntimeError("LZ4F_compressFrame failed")
    return dst.raw[:r]


def lz4_compress_block(src: bytes, cap: int) -> bytes:
    lib = lz4lib()
    dst = ctypes.create_string_buffer(max(cap, 1))
    r = lib.LZ4_compress_default(src, dst, len(src), cap)
    return dst.raw[:r]
