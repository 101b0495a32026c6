"""Multi-GPU sharding of one logical stream: chunks are independent, so rank g takes a contiguous
range of whole hw_buff_sz chunks and the only exchange is a small record per rank (raw bytes, compressed bytes, CRC-32 of
the shard) from which every rank derives its output offset and rank 0 folds the gzip trailer, plus the flat gather of the
compressed shards into rank 0's HBM (OneStream: IPC window or RCCL).  The host arithmetic is covered on CPU by
tests/test_dist_gloo.py; checking a member against the oracle is the callers' business (tests/, bench.py) - nothing in
this package loads the oracle."""
import struct


def shard_chunks(nchunks: int, world: int, rank: int):
    """contiguous [begin, end) chunk range of `rank` (sizes differ by at most one chunk)"""
    q, r = divmod(nchunks, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def _gf2_times(mat, vec):
    s = 0
    i = 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[i]) for i in range(32)]


def crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """zlib crc32_combine(): CRC of A||B from crc(A), crc(B), len(B)"""
    if len2 == 0:
        return crc1
    odd = [0xEDB88320] + [1 << (i - 1) for i in range(1, 32)]
    even = _gf2_square(odd)
    odd = _gf2_square(even)
    while True:
        even = _gf2_square(odd)
        if len2 & 1:
            crc1 = _gf2_times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = _gf2_square(even)
        if len2 & 1:
            crc1 = _gf2_times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def pack_record(raw_len: int, comp_len: int, crc: int) -> bytes:
    return struct.pack("<QII", raw_len, comp_len, crc)


def fold_records(records):
    """records: per-rank (raw_len, comp_len, crc) in rank order -> (offsets, total_raw, total_comp, crc of the whole)"""
    offs, raw, comp, crc = [], 0, 0, 0
    for i, (rl, cl, c) in enumerate(records):
        offs.append(comp)
        crc = c if i == 0 else crc32_combine(crc, c, rl)
        raw += rl
        comp += cl
    return offs, raw, comp, crc


def all_gather_records(pg, rank_record: bytes, world: int):
    """16 B/rank exchange over a torch.distributed process group (gloo on CPU tensors)"""
    import torch
    t = torch.frombuffer(bytearray(rank_record), dtype=torch.uint8).clone()
    outs = [torch.zeros(16, dtype=torch.uint8) for _ in range(world)]
    pg.all_gather(outs, t)
    return [struct.unpack("<QII", bytes(o.tolist())) for o in outs]


def broadcast_bytes(pg, data: bytes, nbytes: int, src: int = 0) -> bytes:
    """a small blob from rank `src` to everybody (gloo, CPU tensors): how the root's IPC handle travels"""
    import torch
    t = torch.zeros(nbytes, dtype=torch.uint8)
    if data is not None:
        t[:len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
    pg.broadcast(t, src)
    return bytes(t.tolist())


def _all_ok(pg, ok: bool) -> bool:
    """every rank learns whether every rank is fine (so that nobody waits for a rank that gave up)"""
    if pg is None:
        return ok
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    pg.all_reduce(t, op=pg.ReduceOp.MIN)
    return bool(t[0])


class OneStream:
    """ONE gzip-ext member out of `world` shards (rank r holds shard r): every rank deflates its shard on its GPU, the
    compressed shards travel to rank 0's HBM, rank 0 folds the CRCs and closes the member (BASELINE config 5's shape; the
    reference's in-order retire across accelerators, src/qatzip.c:1691-1718).  Two transports, same bytes:
      "ipc"   peer copies into a window in rank 0's HBM that every rank maps through HIP IPC (qzd_shard_*)
      "rccl"  one ncclAllGather of the 32-byte records + one ncclSend/ncclRecv group (qzd_rccl_*)
    The object keeps its window / communicator for as many members as the caller builds (run()).  Every step that can
    fail on one rank only is followed by an agreement among the ranks (pg: a torch.distributed-like group, gloo), so a
    failure ends with self.error set everywhere instead of ranks waiting for each other."""

    def __init__(self, ctx, pg, rank, world, shard_bytes, chunk, level=1, transport="ipc", timeout_s=60.0):
        import ctypes as C
        from . import _lib
        self.ctx, self.pg, self.rank, self.world = ctx, pg, rank, world
        self.n, self.chunk, self.level, self.transport, self.timeout_s = shard_bytes, chunk, level, transport, timeout_s
        self.L = L = ctx.L
        self.h = C.c_void_p()
        self.error = None
        self.seq = 0
        self.cap = world * (_lib.max_deflate_len(shard_bytes, chunk) + 64)
        err = None
        if transport == "rccl":
            idb = C.create_string_buffer(128)
            if rank == 0 and L.qzd_rccl_unique_id(idb) != 0:
                err = "ncclGetUniqueId failed (librccl.so.1 missing?)"
            uid = broadcast_bytes(pg, idb.raw if rank == 0 else None, 128, 0) if world > 1 else idb.raw
            if not _all_ok(pg, err is None):
                self.error = err or "rank 0 could not make an RCCL id"
                return
            if L.qzd_rccl_create(ctx.h, rank, world, uid, self.cap, C.byref(self.h)) != 0:
                err = "rccl init: " + L.qzd_last_error(ctx.h).decode()
        else:
            hbuf = C.create_string_buffer(64)
            if rank == 0 and L.qzd_shard_root_create(ctx.h, world, self.cap, hbuf, C.byref(self.h)) != 0:
                err = "root window: " + L.qzd_last_error(ctx.h).decode()
            handle = broadcast_bytes(pg, hbuf.raw if rank == 0 else None, 64, 0) if world > 1 else hbuf.raw
            if rank != 0 and err is None and L.qzd_shard_attach(ctx.h, rank, world, handle, self.cap, C.byref(self.h)) != 0:
                err = "attach: " + L.qzd_last_error(ctx.h).decode()
        if not _all_ok(pg, err is None):
            self.error = err or "another rank could not set the %s transport up" % transport
            self.close()
            return
        self.d_comp = ctx.alloc(_lib.max_deflate_len(shard_bytes, chunk))

    def run(self, d_src, want_member=False):
        """one member.  Rank 0 gets {"raw_bytes", "member_bytes", "ms", "deflate_ms", "gather_ms", "crc32", ("stream")};
        the other ranks {"ms", ...}; {"error": ...} everywhere when any rank failed."""
        import ctypes as C
        import time
        import numpy as np
        L, ctx, rank, world, n, chunk = self.L, self.ctx, self.rank, self.world, self.n, self.chunk
        self.seq += 1
        err = None
        out = {}
        if self.pg is not None:
            self.pg.barrier()
        t0 = time.perf_counter()
        t1 = t0
        clen = crc = 0
        try:
            clen, crcs = ctx.deflate_raw(d_src, n, chunk, self.level, 1 if rank == world - 1 else 0, self.d_comp)
            crcs = np.ascontiguousarray(crcs, dtype=np.uint32)  # the shard's CRC-32 from its chunks' (crc32_combine algebra)
            crc = L.qzd_crc32_fold(crcs.ctypes.data, len(crcs), chunk, n)
            t1 = time.perf_counter()
            dptr, slen, fcrc, raw = C.c_void_p(), C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
            if self.transport == "rccl":
                if L.qzd_rccl_gather(self.h, self.d_comp.ptr, clen, n, crc, self.level, C.byref(dptr), C.byref(slen),
                                     C.byref(fcrc), C.byref(raw)) != 0:
                    err = "gather: " + L.qzd_last_error(ctx.h).decode()
            else:
                if L.qzd_shard_put(self.h, self.d_comp.ptr, clen, n, crc, self.seq, self.timeout_s, None) != 0:
                    err = "put: " + L.qzd_last_error(ctx.h).decode()
                elif rank == 0 and L.qzd_shard_finish(self.h, self.seq, self.timeout_s, self.level, C.byref(dptr), C.byref(slen),
                                                      C.byref(fcrc), C.byref(raw)) != 0:
                    err = "finish: " + L.qzd_last_error(ctx.h).decode()
        except Exception as e:   # noqa: BLE001
            err = "deflate: " + str(e)[:150]
        t2 = time.perf_counter()
        if not _all_ok(self.pg, err is None):
            return {"error": err or "another rank failed"}
        out = {"ms": (t2 - t0) * 1e3, "deflate_ms": (t1 - t0) * 1e3, "gather_ms": (t2 - t1) * 1e3, "comp_len": clen, "crc": crc}
        if rank == 0:
            out.update({"raw_bytes": raw.value, "member_bytes": slen.value, "crc32": "%08x" % fcrc.value})
            if want_member:
                member = np.empty(slen.value, np.uint8)
                ctx._chk(L.qzd_d2h(ctx.h, member.ctypes.data, dptr.value, member.size))
                out["stream"] = member.tobytes()
        if self.pg is not None:
            self.pg.barrier()                                   # nobody starts the next member before rank 0 has read this one
        return out

    def close(self):
        if self.h:
            (self.L.qzd_rccl_close if self.transport == "rccl" else self.L.qzd_shard_close)(self.h)
            self.h = None
        if getattr(self, "d_comp", None) is not None:
            self.d_comp.free(); self.d_comp = None


def member_is_consistent(member: bytes, records) -> bool:
    """pure arithmetic on a finished member: the gzip-ext header's two sizes and the trailer against the ranks' records
    (raw_len, comp_len, crc32 of the shard's plain bytes) - no decompression, no oracle"""
    _, raw_t, comp_t, crc_t = fold_records(records)
    return len(member) == 24 + comp_t + 8 and member[:4] == b"\x1f\x8b\x08\x04" and member[12:14] == b"QZ" and \
        int.from_bytes(member[16:20], "little") == raw_t and int.from_bytes(member[20:24], "little") == comp_t and \
        int.from_bytes(member[-8:-4], "little") == crc_t and int.from_bytes(member[-4:], "little") == (raw_t & 0xffffffff)


def one_stream(ctx, pg, rank, world, d_src, n, chunk, level=1, timeout_s=30.0, transport="ipc", want_member=False):
    """one member, window / communicator made and dropped around it (tests; bench.py keeps a OneStream for its timed loop)"""
    os_ = OneStream(ctx, pg, rank, world, n, chunk, level, transport, timeout_s)
    if os_.error:
        return {"error": os_.error}
    try:
        return os_.run(d_src, want_member)
    finally:
        os_.close()


def allreduce(pg, v: float, op: str) -> float:
    if pg is None:
        return v
    import torch
    t = torch.tensor([v], dtype=torch.float64)
    pg.all_reduce(t, op=getattr(pg.ReduceOp, op))
    return float(t[0])
