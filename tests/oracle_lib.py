"""ctypes loader for oracle/libqzoracle.so (TEST INFRASTRUCTURE - see oracle/qzo.h)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
FMT = {"4B": 0, "GZIP": 1, "GZIP_EXT": 2, "RAW": 3, "LZ4": 4, "ZLIB": 6}
_lib = None


def build():
    so = os.path.join(ODIR, "libqzoracle.so")
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ODIR])
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        u8p = C.c_char_p
        L.qzo_crc32.argtypes = [C.c_uint, u8p, C.c_size_t]; L.qzo_crc32.restype = C.c_uint
        L.qzo_crc32_combine.argtypes = [C.c_uint, C.c_uint, C.c_ulonglong]; L.qzo_crc32_combine.restype = C.c_uint
        L.qzo_adler32.argtypes = [C.c_uint, u8p, C.c_size_t]; L.qzo_adler32.restype = C.c_uint
        L.qzo_xxh32.argtypes = [u8p, C.c_size_t, C.c_uint]; L.qzo_xxh32.restype = C.c_uint
        L.qzo_deflate_chunk.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.qzo_deflate_chunk.restype = C.c_size_t
        L.qzo_deflate_symbols.argtypes = [u8p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.qzo_deflate_symbols.restype = C.c_size_t
        L.qzo_sw_compress.argtypes = [C.c_int, C.c_int, C.c_uint, u8p, C.POINTER(C.c_uint), C.c_void_p,
                                      C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_ulong)]
        L.qzo_sw_decompress.argtypes = [C.c_int, u8p, C.POINTER(C.c_uint), C.c_void_p, C.POINTER(C.c_uint)]
        L.qzo_inflate_raw.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_size_t)]
        L.qzo_lz4_compress_block.argtypes = [u8p, C.c_int, C.c_void_p, C.c_int]
        L.qzo_lz4_decompress_block.argtypes = [u8p, C.c_int, C.c_void_p, C.c_int]
        L.qzo_lz4f_compress_frame.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.qzo_lz4f_compress_frame.restype = C.c_size_t
        L.qzo_lz4f_bound.argtypes = [C.c_size_t]; L.qzo_lz4f_bound.restype = C.c_size_t
        _lib = L
    return _lib


def sw_compress(fmt, src: bytes, hw=65536, level=1, last=1, crc0=0, cap=None):
    """-> (rc, consumed, out_bytes, crc)"""
    L = lib()
    if cap is None:
        cap = len(src) * 9 // 8 + 2048
    sl = C.c_uint(len(src)); dl = C.c_uint(cap); dst = C.create_string_buffer(max(cap, 1)); crc = C.c_ulong(crc0)
    rc = L.qzo_sw_compress(FMT[fmt] if isinstance(fmt, str) else fmt, level, hw, src, C.byref(sl), dst,
                           C.byref(dl), last, C.byref(crc))
    return rc, sl.value, dst.raw[:dl.value], crc.value


def sw_decompress(fmt, comp: bytes, cap: int):
    """-> (rc, consumed, out_bytes)"""
    L = lib()
    sl = C.c_uint(len(comp)); dl = C.c_uint(cap); dst = C.create_string_buffer(max(cap, 1))
    rc = L.qzo_sw_decompress(FMT[fmt] if isinstance(fmt, str) else fmt, comp, C.byref(sl), dst, C.byref(dl))
    return rc, sl.value, dst.raw[:dl.value]


def deflate_chunk(src: bytes, level=1, final=1):
    L = lib()
    cap = len(src) * 9 // 8 + 2048
    dst = C.create_string_buffer(cap)
    r = L.qzo_deflate_chunk(src, len(src), dst, cap, level, final)
    assert r != C.c_size_t(-1).value
    return dst.raw[:r]


def deflate_symbols(src: bytes, level=1):
    import numpy as np
    L = lib()
    cap = len(src) + 8
    lc = np.zeros(cap, np.uint8); dist = np.zeros(cap, np.uint16)
    n = L.qzo_deflate_symbols(src, len(src), level, lc.ctypes.data, dist.ctypes.data, cap)
    return lc[:n].copy(), dist[:n].copy()


def lz4_compress_block(src: bytes, cap=None):
    L = lib()
    if cap is None:
        cap = len(src) - 1
    dst = C.create_string_buffer(max(cap, 1))
    r = L.qzo_lz4_compress_block(src, len(src), dst, cap)
    return dst.raw[:r]
