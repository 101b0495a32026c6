"""Multi-GPU sharding of one logical stream: chunks are independent, so rank g takes a contiguous
range of whole hw_buff_sz chunks and the only exchange is a 16-byte record per rank
(raw bytes, compressed bytes, CRC-32 of the shard) from which every rank derives its output offset and rank 0
folds the gzip trailer.  No data-path collective: the compressed shards are written at their offsets
(or gathered by the caller).  Pure host logic - used by bench.py and covered on CPU by tests/test_dist_gloo.py."""
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


def one_stream(ctx, pg, rank, world, d_src, n, chunk, level=1, timeout_s=30.0, seq=1, verify=None):
    """ONE gzip-ext member out of `world` shards of n bytes each (rank r holds shard r in d_src): every rank deflates its
    shard on its GPU, the compressed shards travel to rank 0's HBM as peer copies (qzd_shard_*), rank 0 closes the member.
    Returns a dict (rank 0: sizes, time, the member's bytes under "stream" if verify == "full"); other ranks: {}.
    verify: None | "sample" (rank 0's own shard prefix against the oracle + trailer against the ranks' CPU CRCs) | "full".
    Every step that can fail on one rank only (IPC mapping, peer copy) is followed by an agreement among the ranks, so a
    failure ends the leg everywhere with {"error": ...} instead of leaving ranks waiting for each other."""
    import ctypes as C
    import time
    import zlib
    import numpy as np
    from . import _lib
    L = ctx.L
    cap = world * (_lib.max_deflate_len(n, chunk) + 64)
    hbuf = C.create_string_buffer(64)
    win = C.c_void_p()
    err = None
    if rank == 0:
        if L.qzd_shard_root_create(ctx.h, world, cap, hbuf, C.byref(win)) != 0:
            err = "root window: " + L.qzd_last_error(ctx.h).decode()
    handle = broadcast_bytes(pg, hbuf.raw if rank == 0 else None, 64, 0) if world > 1 else hbuf.raw
    if rank != 0 and err is None:
        if L.qzd_shard_attach(ctx.h, rank, world, handle, cap, C.byref(win)) != 0:
            err = "attach: " + L.qzd_last_error(ctx.h).decode()
    if not _all_ok(pg, err is None):
        if win:
            L.qzd_shard_close(win)
        return {"error": err or "another rank could not map the window"}
    d_comp = ctx.alloc(_lib.max_deflate_len(n, chunk))
    if pg is not None:
        pg.barrier()
    t0 = time.perf_counter()
    clen, crc = 0, 0
    try:
        clen, crcs = ctx.deflate_raw(d_src, n, chunk, level, 1 if rank == world - 1 else 0, d_comp)
        for i, c in enumerate(crcs):                            # the shard's CRC-32 from its chunks' (crc32_combine algebra)
            cl = min(chunk, n - i * chunk)
            crc = int(c) if i == 0 else L.qzd_crc32_combine(crc, int(c), cl)
        off = C.c_uint64(0)
        if L.qzd_shard_put(win, d_comp.ptr, clen, n, crc, seq, timeout_s, C.byref(off)) != 0:
            err = "put: " + L.qzd_last_error(ctx.h).decode()
    except Exception as e:   # noqa: BLE001
        err = "deflate: " + str(e)[:150]
    out = {}
    dptr = C.c_void_p()
    if rank == 0 and err is None:
        slen, fcrc, raw = C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
        if L.qzd_shard_finish(win, seq, timeout_s, C.byref(dptr), C.byref(slen), C.byref(fcrc), C.byref(raw)) != 0:
            err = "finish: " + L.qzd_last_error(ctx.h).decode()
        else:
            dt = time.perf_counter() - t0
            out = {"ranks": world, "raw_bytes": raw.value, "member_bytes": slen.value, "ms": round(dt * 1e3, 2),
                   "GBps": round(raw.value / dt / 1e9, 2), "crc32": "%08x" % fcrc.value,
                   "transport": "peer copies into an IPC window in rank 0's HBM (xGMI between GPUs)"}
    if not _all_ok(pg, err is None):
        L.qzd_shard_close(win)
        d_comp.free()
        return {"error": err or "another rank failed"}
    if verify:
        host = d_src.download(n)
        my_crc = zlib.crc32(host.tobytes()) & 0xffffffff
        recs = all_gather_records(pg, pack_record(n, clen, my_crc), world) if world > 1 else [(n, clen, my_crc)]
        if rank == 0:
            _, raw_t, comp_t, crc_t = fold_records(recs)
            member = np.empty(out["member_bytes"], np.uint8)
            ctx._chk(L.qzd_d2h(ctx.h, member.ctypes.data, dptr.value, member.size))
            mb = member.tobytes()
            ok = mb[:4] == b"\x1f\x8b\x08\x04" and int.from_bytes(mb[16:20], "little") == raw_t and \
                int.from_bytes(mb[20:24], "little") == comp_t and int.from_bytes(mb[-8:-4], "little") == crc_t and \
                int.from_bytes(mb[-4:], "little") == (raw_t & 0xffffffff) and int(out["crc32"], 16) == crc_t
            if ok:
                import sys
                import os
                sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
                import oracle_lib as O
                k = min(n, 4 << 20) // chunk * chunk
                if k:
                    exp = O.sw_compress("RAW", host[:k].tobytes(), chunk, level, last=0 if (world > 1 or k < n) else 1, cap=k * 9 // 8 + 65536)[2]
                    ok = mb[24:24 + len(exp)] == exp
            out["verified"] = bool(ok)
            if verify == "full":
                out["stream"] = mb
    if pg is not None:
        pg.barrier()                                            # nobody unmaps the window before rank 0 has read it
    L.qzd_shard_close(win)
    d_comp.free()
    return out


def allreduce(pg, v: float, op: str) -> float:
    if pg is None:
        return v
    import torch
    t = torch.tensor([v], dtype=torch.float64)
    pg.all_reduce(t, op=getattr(pg.ReduceOp, op))
    return float(t[0])
