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
        if libz_pinned():
            # the whole member - header with its 'Q','Z' extra field included - written by libz itself through its C API, driven
            # as qzDeflateSWCompress drives it (libz_gzip_ext below); the typed-out header that follows is what it is equal to
            # (tests/test_oracle.py::test_gzip_ext_members_written_by_libz_itself) and what runs where libz.so.1 is missing
            return libz_gzip_ext(src, hw, level, last)
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


# ---------------------------------------------------------------- libz itself, through its C API (ctypes)
# The functions above go through CPython's zlib module, which cannot hand deflate() a gzip header of the caller's: the 24
# bytes in front of a GZIP_EXT stream were typed out here.  This drives the system libz exactly as qzDeflateSWCompress does
# (src/qatzip_sw.c:61-75, 147-166, 178-253): deflateInit2_(level, Z_DEFLATED, 15 + 16, MAX_MEM_LEVEL, Z_DEFAULT_STRATEGY),
# deflateSetHeader() with the 12-byte 'Q','Z' extra field and os = 255, one deflate(Z_FULL_FLUSH) per hw_buff_sz chunk
# into the caller's whole destination, deflate(Z_FINISH) on the last, then the two size fields patched into the header
# zlib wrote - so that zlib writes every byte of the member, header included.
class _ZStream(ctypes.Structure):
    _fields_ = [("next_in", ctypes.c_void_p), ("avail_in", ctypes.c_uint), ("total_in", ctypes.c_ulong),
                ("next_out", ctypes.c_void_p), ("avail_out", ctypes.c_uint), ("total_out", ctypes.c_ulong),
                ("msg", ctypes.c_char_p), ("state", ctypes.c_void_p), ("zalloc", ctypes.c_void_p), ("zfree", ctypes.c_void_p),
                ("opaque", ctypes.c_void_p), ("data_type", ctypes.c_int), ("adler", ctypes.c_ulong), ("reserved", ctypes.c_ulong)]


class _GzHeader(ctypes.Structure):
    _fields_ = [("text", ctypes.c_int), ("time", ctypes.c_ulong), ("xflags", ctypes.c_int), ("os", ctypes.c_int),
                ("extra", ctypes.c_void_p), ("extra_len", ctypes.c_uint), ("extra_max", ctypes.c_uint),
                ("name", ctypes.c_void_p), ("name_max", ctypes.c_uint), ("comment", ctypes.c_void_p), ("comm_max", ctypes.c_uint),
                ("hcrc", ctypes.c_int), ("done", ctypes.c_int)]


_libz = None


def libz():
    global _libz
    if _libz is None:
        try:
            lib = ctypes.CDLL("libz.so.1")
            lib.zlibVersion.restype = ctypes.c_char_p
            lib.deflateInit2_.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_char_p, ctypes.c_int]
            lib.deflateSetHeader.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            lib.deflate.argtypes = [ctypes.c_void_p, ctypes.c_int]
            lib.deflateEnd.argtypes = [ctypes.c_void_p]
            _libz = lib
        except OSError:
            _libz = False
    return _libz or None


def libz_pinned():
    lib = libz()
    return bool(lib) and lib.zlibVersion() == b"1.2.11"


def libz_gzip_ext(src: bytes, hw: int = 65536, level: int = 1, last: int = 1) -> bytes:
    """one qzDeflateSWCompress(DEFLATE_GZIP_EXT) call: every byte, the header included, written by libz"""
    lib = libz()
    Z_DEFLATED, Z_FULL_FLUSH, Z_FINISH, Z_OK, Z_STREAM_END = 8, 3, 4, 0, 1
    strm = _ZStream()
    ver = lib.zlibVersion()
    assert lib.deflateInit2_(ctypes.byref(strm), level, Z_DEFLATED, 15 + 16, 9, 0, ver, ctypes.sizeof(_ZStream)) == Z_OK
    extra = ctypes.create_string_buffer(b"QZ" + struct.pack("<HII", 8, 0, 0), 12)      # g_extra_field: 'Q','Z', x2_len 8, src_sz 0, dest_sz 0
    hdr = _GzHeader()                                                                 # gen_qatzip_hdr: zeroed, extra, os = 255
    hdr.extra = ctypes.cast(extra, ctypes.c_void_p); hdr.extra_len = 12; hdr.os = 255
    assert lib.deflateSetHeader(ctypes.byref(strm), ctypes.byref(hdr)) == Z_OK
    cap = len(src) * 9 // 8 + 1024 * (len(src) // hw + 2)
    dst = ctypes.create_string_buffer(cap)
    inb = ctypes.create_string_buffer(src, max(len(src), 1))
    pos, n = 0, len(src)
    while True:
        send = min(hw, n - pos)
        strm.next_in = ctypes.cast(ctypes.byref(inb, pos), ctypes.c_void_p); strm.avail_in = send
        strm.next_out = ctypes.cast(ctypes.byref(dst, strm.total_out), ctypes.c_void_p); strm.avail_out = cap - strm.total_out
        pos += send
        fin = pos == n and last == 1
        rc = lib.deflate(ctypes.byref(strm), Z_FINISH if fin else Z_FULL_FLUSH)
        assert rc == (Z_STREAM_END if fin else Z_OK) and strm.avail_in == 0, rc
        if pos == n:
            break
    out = bytearray(dst.raw[:strm.total_out])
    if last == 1:                                                                     # src/qatzip_sw.c:238-243
        out[16:20] = struct.pack("<I", strm.total_in & 0xffffffff)
        out[20:24] = struct.pack("<I", (strm.total_out - 24 - 8) & 0xffffffff)
    lib.deflateEnd(ctypes.byref(strm))
    return bytes(out)
