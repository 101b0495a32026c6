"""ctypes mirror of the qatzip.h application interface exported by libqatzip_amd.so.

Same names, argument meaning and return codes as the reference API (include/qatzip.h), so the parity
tests read like the reference's own tests (test/main.c, test/bt.c)."""
import ctypes as C

from ._lib import load

QZ_OK, QZ_DUPLICATE, QZ_PARAMS, QZ_FAIL, QZ_BUF_ERROR, QZ_DATA_ERROR = 0, 1, -1, -2, -3, -4
QZ_NOT_SUPPORTED, QZ_NOSW_NO_HW, QZ_UNSUPPORTED_FMT = -200, -101, 16
QZ_DEFLATE_4B, QZ_DEFLATE_GZIP, QZ_DEFLATE_GZIP_EXT, QZ_DEFLATE_RAW = 0, 1, 2, 3
QZ_DEFLATE, QZ_LZ4 = 8, ord("4")
QZ_DIR_COMPRESS, QZ_DIR_DECOMPRESS, QZ_DIR_BOTH = 0, 1, 2
COMMON_MEM, PINNED_MEM = 0, 1


class QzSession(C.Structure):
    _fields_ = [("hw_session_stat", C.c_long), ("thd_sess_stat", C.c_int), ("internal", C.c_void_p),
                ("total_in", C.c_ulong), ("total_out", C.c_ulong)]


class QzSessionParams(C.Structure):
    _fields_ = [("huffman_hdr", C.c_int), ("direction", C.c_int), ("data_fmt", C.c_int), ("comp_lvl", C.c_uint),
                ("comp_algorithm", C.c_ubyte), ("max_forks", C.c_uint), ("sw_backup", C.c_ubyte),
                ("hw_buff_sz", C.c_uint), ("strm_buff_sz", C.c_uint), ("input_sz_thrshold", C.c_uint),
                ("req_cnt_thrshold", C.c_uint), ("wait_cnt_thrshold", C.c_uint)]


class QzSessionParamsCommon(C.Structure):
    _fields_ = [("direction", C.c_int), ("comp_lvl", C.c_uint), ("comp_algorithm", C.c_ubyte), ("max_forks", C.c_uint),
                ("sw_backup", C.c_ubyte), ("hw_buff_sz", C.c_uint), ("strm_buff_sz", C.c_uint),
                ("input_sz_thrshold", C.c_uint), ("req_cnt_thrshold", C.c_uint), ("wait_cnt_thrshold", C.c_uint),
                ("polling_mode", C.c_int), ("is_sensitive_mode", C.c_uint)]


class QzSessionParamsDeflate(C.Structure):
    _fields_ = [("common_params", QzSessionParamsCommon), ("huffman_hdr", C.c_int), ("data_fmt", C.c_int)]


class QzSessionParamsDeflateExt(C.Structure):
    _fields_ = [("deflate_params", QzSessionParamsDeflate), ("stop_decompression_stream_end", C.c_ubyte),
                ("zlib_format", C.c_ubyte)]


class QzSessionParamsLZ4(C.Structure):
    _fields_ = [("common_params", QzSessionParamsCommon)]


class QzStream(C.Structure):
    _fields_ = [("in_sz", C.c_uint), ("out_sz", C.c_uint), ("in_", C.c_void_p), ("out", C.c_void_p),
                ("pending_in", C.c_uint), ("pending_out", C.c_uint), ("crc_type", C.c_int), ("crc_32", C.c_uint),
                ("reserved", C.c_ulonglong), ("opaque", C.c_void_p)]


class QzResult(C.Structure):
    _fields_ = [("status", C.c_int), ("cb_tag", C.c_void_p), ("src_len", C.c_uint), ("dest_len", C.c_uint),
                ("ext_rc", C.c_uint64), ("crc", C.c_void_p), ("extension_result", C.c_void_p)]


QzAsyncCallback = C.CFUNCTYPE(C.c_int, C.POINTER(QzResult))

_bound = False


def lib():
    global _bound
    L = load()
    if not _bound:
        P, u8p, up = C.POINTER, C.c_char_p, C.POINTER(C.c_uint)
        L.qzInit.argtypes = [P(QzSession), C.c_ubyte]
        L.qzSetupSession.argtypes = [P(QzSession), P(QzSessionParams)]
        L.qzSetupSessionDeflate.argtypes = [P(QzSession), P(QzSessionParamsDeflate)]
        L.qzSetupSessionLZ4.argtypes = [P(QzSession), P(QzSessionParamsLZ4)]
        L.qzGetDefaults.argtypes = [P(QzSessionParams)]
        L.qzSetDefaults.argtypes = [P(QzSessionParams)]
        L.qzGetDefaultsDeflate.argtypes = [P(QzSessionParamsDeflate)]
        L.qzGetDefaultsLZ4.argtypes = [P(QzSessionParamsLZ4)]
        L.qzGetDefaultsDeflateExt.argtypes = [P(QzSessionParamsDeflateExt)]
        L.qzSetupSessionDeflateExt.argtypes = [P(QzSession), P(QzSessionParamsDeflateExt)]
        L.qzCompress.argtypes = [P(QzSession), u8p, up, C.c_void_p, up, C.c_uint]
        L.qzCompressCrc.argtypes = [P(QzSession), u8p, up, C.c_void_p, up, C.c_uint, P(C.c_ulong)]
        L.qzDecompress.argtypes = [P(QzSession), u8p, up, C.c_void_p, up]
        L.qzDecompressCrc.argtypes = [P(QzSession), u8p, up, C.c_void_p, up, P(C.c_ulong)]
        L.qzTeardownSession.argtypes = [P(QzSession)]
        L.qzClose.argtypes = [P(QzSession)]
        L.qzMaxCompressedLength.argtypes = [C.c_uint, P(QzSession)]
        L.qzMaxCompressedLength.restype = C.c_uint
        L.qzMalloc.argtypes = [C.c_size_t, C.c_int, C.c_int]; L.qzMalloc.restype = C.c_void_p
        L.qzFree.argtypes = [C.c_void_p]; L.qzFree.restype = None
        L.qzMemFindAddr.argtypes = [C.c_void_p]
        L.qzCompressStream.argtypes = [P(QzSession), P(QzStream), C.c_uint]
        L.qzDecompressStream.argtypes = [P(QzSession), P(QzStream), C.c_uint]
        L.qzEndStream.argtypes = [P(QzSession), P(QzStream)]
        L.qzSetLogLevel.argtypes = [C.c_int]
        L.qzCompress2.argtypes = [P(QzSession), C.c_void_p, C.c_void_p, C.c_void_p, P(QzResult)]
        L.qzDecompress2.argtypes = [P(QzSession), C.c_void_p, C.c_void_p, C.c_void_p, P(QzResult)]
        _bound = True
    return L


class Session:
    """A QzSession_T set up the way test/main.c does it: qzGetDefaults -> tweak -> qzSetupSession."""

    def __init__(self, data_fmt=QZ_DEFLATE_GZIP_EXT, hw_buff_sz=65536, comp_lvl=1, lz4=False, strm_buff_sz=None,
                 zlib_format=False, stop_at_stream_end=False):
        self.L = lib()
        self.s = QzSession()
        if zlib_format or stop_at_stream_end:  # qzSetupSessionDeflateExt: zlib_format = 1 is the RFC 1950 wrapper with an Adler-32 trailer
            p = QzSessionParamsDeflateExt(); self.L.qzGetDefaultsDeflateExt(C.byref(p))
            if zlib_format:
                p.deflate_params.data_fmt = QZ_DEFLATE_RAW; p.zlib_format = 1
            else:
                p.deflate_params.data_fmt = data_fmt
            p.stop_decompression_stream_end = 1 if stop_at_stream_end else 0
            p.deflate_params.common_params.hw_buff_sz = hw_buff_sz; p.deflate_params.common_params.comp_lvl = comp_lvl
            self.rc_setup = self.L.qzSetupSessionDeflateExt(C.byref(self.s), C.byref(p))
        elif lz4:
            p = QzSessionParamsLZ4(); self.L.qzGetDefaultsLZ4(C.byref(p))
            p.common_params.comp_algorithm = QZ_LZ4
            p.common_params.hw_buff_sz = hw_buff_sz; p.common_params.comp_lvl = comp_lvl
            self.rc_setup = self.L.qzSetupSessionLZ4(C.byref(self.s), C.byref(p))
        else:
            p = QzSessionParams(); self.L.qzGetDefaults(C.byref(p))
            p.data_fmt = data_fmt; p.hw_buff_sz = hw_buff_sz; p.comp_lvl = comp_lvl
            if strm_buff_sz:
                p.strm_buff_sz = strm_buff_sz
            self.rc_setup = self.L.qzSetupSession(C.byref(self.s), C.byref(p))

    def compress(self, src: bytes, last=1, cap=None, crc0=None):
        """-> (rc, consumed, out_bytes, crc)"""
        if cap is None:
            cap = self.L.qzMaxCompressedLength(max(len(src), 1), C.byref(self.s)) + 64
        sl, dl = C.c_uint(len(src)), C.c_uint(cap)
        dst = C.create_string_buffer(max(cap, 1))
        if crc0 is None:
            rc = self.L.qzCompress(C.byref(self.s), src, C.byref(sl), dst, C.byref(dl), last)
            return rc, sl.value, dst.raw[:dl.value], None
        crc = C.c_ulong(crc0)
        rc = self.L.qzCompressCrc(C.byref(self.s), src, C.byref(sl), dst, C.byref(dl), last, C.byref(crc))
        return rc, sl.value, dst.raw[:dl.value], crc.value

    def decompress(self, comp: bytes, cap: int, crc0=0, want_crc=False):
        """-> (rc, consumed, out_bytes[, crc]): qzDecompress, or qzDecompressCrc (running CRC-32 of the output) with want_crc"""
        sl, dl = C.c_uint(len(comp)), C.c_uint(cap)
        dst = C.create_string_buffer(max(cap, 1))
        if want_crc:
            crc = C.c_ulong(crc0)
            rc = self.L.qzDecompressCrc(C.byref(self.s), comp, C.byref(sl), dst, C.byref(dl), C.byref(crc))
            return rc, sl.value, dst.raw[:dl.value], crc.value
        rc = self.L.qzDecompress(C.byref(self.s), comp, C.byref(sl), dst, C.byref(dl))
        return rc, sl.value, dst.raw[:dl.value]

    def end_of_stream(self):
        """qzGetDeflateEndOfStream: 1 if the last decompress call ended on the end of a deflate stream"""
        e = C.c_ubyte(7)
        rc = self.L.qzGetDeflateEndOfStream(C.byref(self.s), C.byref(e))
        return rc, e.value

    def close(self):
        self.L.qzTeardownSession(C.byref(self.s))
        self.L.qzClose(C.byref(self.s))
