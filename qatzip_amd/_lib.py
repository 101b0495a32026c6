"""ctypes view of libqatzip_amd.so's device-resident C ABI (include/qzamd_device.h).

There is deliberately no fallback: if the HIP library is missing or no GPU is present
the constructors raise.  (The CPU oracle lives under oracle/ and is test-only.)
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


class QzdError(RuntimeError):
    pass


def load(build_if_missing=True):
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("QATZIP_AMD_SO") or _build.SO      # developer switch: a variant build of the same library
    if not os.path.exists(so):
        if not build_if_missing:
            raise QzdError("libqatzip_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        _build.build()
    L = C.CDLL(so)
    vp, u8p = C.c_void_p, C.c_void_p
    L.qzd_device_count.restype = C.c_int
    L.qzd_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.qzd_destroy.argtypes = [vp]
    L.qzd_last_error.argtypes = [vp]; L.qzd_last_error.restype = C.c_char_p
    L.qzd_batch_chunks.argtypes = [vp]; L.qzd_batch_chunks.restype = C.c_uint32
    L.qzd_k1_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    L.qzd_stream_copy_peak.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.qzd_dev_alloc.argtypes = [vp, C.c_size_t]; L.qzd_dev_alloc.restype = vp
    L.qzd_dev_free.argtypes = [vp, vp]
    L.qzd_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.qzd_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.qzd_deflate_raw.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, u8p, C.c_uint64,
                                  C.POINTER(C.c_uint64), vp]
    L.qzd_deflate_raw_async.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, u8p, C.c_uint64]
    L.qzd_sync.argtypes = [vp]
    L.qzd_result.argtypes = [vp, C.POINTER(C.c_uint64), vp, C.c_uint32]
    L.qzd_last_timing.argtypes = [vp, C.POINTER(C.c_float * 4)]
    L.qzd_inflate_segments.argtypes = [vp, u8p, u8p, vp, C.c_uint32, vp]
    L.qzd_inflate_stream.argtypes = [vp, u8p, C.c_uint64, u8p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.qzd_crc32.argtypes = [vp, u8p, C.c_uint64, C.POINTER(C.c_uint32)]
    L.qzd_crc32_ranges.argtypes = [vp, u8p, vp, C.c_uint32, vp]
    L.qzd_last_inflate_timing.argtypes = [vp, C.POINTER(C.c_float * 4)]
    if hasattr(L, "qzd_inflate_scratch_bytes"):            # (variant libraries built from older sources lack it; the export test does not)
        L.qzd_inflate_scratch_bytes.argtypes = [vp]; L.qzd_inflate_scratch_bytes.restype = C.c_uint64
    L.qzd_d2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.qzd_lz4_compress_frames.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, C.POINTER(C.c_uint64), vp]
    L.qzd_lz4_compress_frames_hw.argtypes = L.qzd_lz4_compress_frames.argtypes
    L.qzd_lz4_compress_linked.argtypes = [vp, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.qzd_lz4_decompress_frames.argtypes = [vp, u8p, u8p, vp, C.c_uint32, vp]
    L.qzd_chunk_lens.argtypes = [vp, vp, C.c_uint32]
    L.qzd_shard_root_create.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_char_p, C.POINTER(vp)]
    L.qzd_shard_attach.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(vp)]
    L.qzd_shard_slot_handle.argtypes = [vp, C.c_uint32, C.c_char_p]
    L.qzd_shard_attach_slot.argtypes = [vp, C.c_char_p]
    L.qzd_shard_put.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_uint64)]
    L.qzd_shard_finish.argtypes = [vp, C.c_uint32, C.c_double, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint64)]
    L.qzd_shard_close.argtypes = [vp]
    L.qzd_rccl_unique_id.argtypes = [C.c_char_p]
    L.qzd_rccl_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(vp)]
    L.qzd_rccl_gather.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.qzd_rccl_close.argtypes = [vp]
    L.qzd_ctx_device.argtypes = [vp]
    L.qzd_pcie_peak.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.qzd_crc32_fold.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64]; L.qzd_crc32_fold.restype = C.c_uint32
    L.qzd_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]; L.qzd_crc32_combine.restype = C.c_uint32
    _lib = L
    return L


SEG_DT = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4"),
                   ("flags", "<u4"), ("pad", "<u4")])
RES_DT = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("nblocks", "<u4")])
LZ4SEG_DT = np.dtype([("in_off", "<u8"), ("out_off", "<u8"), ("in_len", "<u4"), ("out_cap", "<u4")])
LZ4RES_DT = np.dtype([("status", "<i4"), ("in_used", "<u4"), ("out_len", "<u4"), ("pad", "<u4")])


def exported_symbols():
    """Names include/*.h declares that must be exported by the library (checked on CPU too)."""
    return ["qzd_create", "qzd_destroy", "qzd_last_error", "qzd_device_count", "qzd_ctx_device", "qzd_dev_alloc", "qzd_dev_free",
            "qzd_h2d", "qzd_d2h", "qzd_host_alloc_pinned", "qzd_host_free_pinned", "qzd_deflate_raw",
            "qzd_deflate_raw_async", "qzd_sync", "qzd_result", "qzd_last_timing", "qzd_inflate_segments",
            "qzd_inflate_stream", "qzd_crc32", "qzd_crc32_ranges", "qzd_last_inflate_timing", "qzd_inflate_scratch_bytes",
            "qzd_lz4_compress_frames", "qzd_lz4_compress_frames_hw", "qzd_lz4_decompress_frames", "qzd_chunk_lens", "qzd_batch_chunks", "qzd_k1_stats",
            "qzd_adler32_chunks", "qzd_adler32_combine", "qzd_stream_copy_peak", "qzd_deflate_raw_from_host",
            "qzd_deflate_slots", "qzd_inflate_stream_to_host", "qzd_inflate_stream_from_host", "qzamd_async_stats", "qzd_shard_root_create",
            "qzd_shard_attach", "qzd_shard_slot_handle", "qzd_shard_attach_slot", "qzd_lz4_compress_linked", "qzd_shard_put", "qzd_shard_finish", "qzd_shard_close", "qzd_crc32_combine",
            "qzd_crc32_fold", "qzd_pcie_peak", "qzd_rccl_unique_id", "qzd_rccl_create", "qzd_rccl_gather", "qzd_rccl_close"]


class DevBuf:
    """A plain HBM allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = ctx.L.qzd_dev_alloc(ctx.h, self.nbytes + 512)   # slack: kernels may look a few bytes ahead
        if not self.ptr:
            raise QzdError("hipMalloc failed for %d bytes" % nbytes)

    def upload(self, data, offset=0):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
        assert offset + a.size <= self.nbytes
        if a.size:
            a = np.ascontiguousarray(a)
            self.ctx._chk(self.ctx.L.qzd_h2d(self.ctx.h, self.ptr + offset, a.ctypes.data, a.size))

    def download(self, n=None, offset=0):
        n = self.nbytes - offset if n is None else int(n)
        out = np.empty(n, np.uint8)
        if n:
            self.ctx._chk(self.ctx.L.qzd_d2h(self.ctx.h, out.ctypes.data, self.ptr + offset, n))
        return out

    def free(self):
        if self.ptr:
            self.ctx.L.qzd_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """qzd_ctx wrapper: one per (process, GPU)."""

    def __init__(self, device=0):
        self.L = load()
        if self.L.qzd_device_count() <= 0:
            raise QzdError("no HIP device visible: the MI355X backend has no CPU fallback")
        h = C.c_void_p()
        rc = self.L.qzd_create(device, C.byref(h))
        if rc != 0:
            raise QzdError("qzd_create failed rc=%d" % rc)
        self.h = h
        self.device = device

    def _chk(self, rc):
        if rc != 0:
            raise QzdError("rc=%d: %s" % (rc, self.L.qzd_last_error(self.h).decode()))

    def alloc(self, n):
        return DevBuf(self, n)

    def close(self):
        if self.h:
            self.L.qzd_destroy(self.h)
            self.h = None

    # -- deflate
    def deflate_raw(self, d_src, n, chunk_sz=65536, level=1, last=1, d_dst=None, want_crc=True):
        """-> (out_len, crc array or None)"""
        nchunks = max(1, (n + chunk_sz - 1) // chunk_sz)
        out_len = C.c_uint64(0)
        crcs = np.zeros(nchunks, np.uint32) if want_crc else None
        self._chk(self.L.qzd_deflate_raw(self.h, d_src.ptr, n, chunk_sz, level, last, d_dst.ptr, d_dst.nbytes,
                                         C.byref(out_len), crcs.ctypes.data if want_crc else None))
        return out_len.value, crcs

    def deflate_raw_async(self, d_src, n, chunk_sz, level, last, d_dst):
        self._chk(self.L.qzd_deflate_raw_async(self.h, d_src.ptr, n, chunk_sz, level, last, d_dst.ptr, d_dst.nbytes))

    def sync(self):
        self._chk(self.L.qzd_sync(self.h))

    def result(self):
        out_len = C.c_uint64(0)
        self._chk(self.L.qzd_result(self.h, C.byref(out_len), None, 0))
        return out_len.value

    def stream_copy_peak(self, nbytes=1 << 30, iters=3):
        """measured HBM stream-copy rate, GB/s of read + written bytes"""
        g = C.c_double(0)
        self._chk(self.L.qzd_stream_copy_peak(self.h, nbytes, iters, C.byref(g)))
        return g.value

    def pcie_peak(self, nbytes=1 << 30, iters=2):
        """pinned hipMemcpyAsync rates (host -> device, device -> host), GB/s"""
        a, b = C.c_double(0), C.c_double(0)
        self._chk(self.L.qzd_pcie_peak(self.h, nbytes, iters, C.byref(a), C.byref(b)))
        return a.value, b.value

    def batch_chunks(self):
        return int(self.L.qzd_batch_chunks(self.h))

    def k1_stats(self, reset=False):
        """(total ms, launches, chunks) of the LZ77 kernel since the last reset"""
        ms, ln, ch = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        self.L.qzd_k1_stats(self.h, C.byref(ms), C.byref(ln), C.byref(ch), 1 if reset else 0)
        return ms.value, ln.value, ch.value

    def timing(self):
        ms = (C.c_float * 4)()
        self.L.qzd_last_timing(self.h, C.byref(ms))
        return list(ms)

    # -- inflate
    def inflate_segments(self, d_comp, d_out, segs):
        """segs: list of (in_off, out_off, in_len, out_cap, flags) -> structured result array"""
        sa = np.array([tuple(s) + (0,) for s in segs], dtype=SEG_DT)
        res = np.zeros(len(segs), RES_DT)
        self._chk(self.L.qzd_inflate_segments(self.h, d_comp.ptr, d_out.ptr, sa.ctypes.data, len(segs), res.ctypes.data))
        return res

    def inflate_stream(self, d_src, n, d_dst, seg_hint=65536, want_crc=True, src_off=0):
        """-> (in_used, out_len, crc or None); raises QzdError on corrupt data / short destination"""
        iu, ol, crc = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        self._chk(self.L.qzd_inflate_stream(self.h, d_src.ptr + src_off, n, d_dst.ptr, d_dst.nbytes, seg_hint,
                                            C.byref(iu), C.byref(ol), C.byref(crc) if want_crc else None))
        return iu.value, ol.value, (crc.value if want_crc else None)

    def crc32(self, d_data, n):
        crc = C.c_uint32(0)
        self._chk(self.L.qzd_crc32(self.h, d_data.ptr, n, C.byref(crc)))
        return crc.value

    # -- LZ4
    def lz4_compress_frames(self, d_src, n, d_dst, frame_sz=65536):
        """-> (out_len, per-frame lengths)"""
        nfr = max(1, (n + frame_sz - 1) // frame_sz)
        ol = C.c_uint64(0)
        lens = np.zeros(nfr, np.uint32)
        self._chk(self.L.qzd_lz4_compress_frames(self.h, d_src.ptr, n, frame_sz, d_dst.ptr, d_dst.nbytes, C.byref(ol),
                                                 lens.ctypes.data))
        return ol.value, lens

    def lz4_decompress_frames(self, d_comp, d_out, segs):
        """segs: list of (in_off, out_off, in_len, out_cap) -> structured results (status, in_used, out_len)"""
        sa = np.array([tuple(s) for s in segs], dtype=LZ4SEG_DT)
        res = np.zeros(len(segs), LZ4RES_DT)
        self._chk(self.L.qzd_lz4_decompress_frames(self.h, d_comp.ptr, d_out.ptr, sa.ctypes.data, len(segs), res.ctypes.data))
        return res

    def inflate_timing(self):
        ms = (C.c_float * 4)()
        self.L.qzd_last_inflate_timing(self.h, C.byref(ms))
        return list(ms)


def max_deflate_len(n, chunk_sz=65536):
    """worst-case raw stream size for n bytes (stored blocks + markers)"""
    nchunks = max(1, (n + chunk_sz - 1) // chunk_sz)
    return n + nchunks * (5 * (chunk_sz // 32767 + 2) + 16) + 64
