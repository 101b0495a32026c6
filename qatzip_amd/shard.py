"""Multi-GPU sharding of one logical stream: chunks are independent, so rank g takes a contiguous
range of whole hw_buff_sz chunks and the only exchange is a small record per rank (raw bytes, compressed bytes, CRC-32 of
the shard) from which every rank derives its output offset and rank 0 folds the gzip trailer, plus the flat gather of the
compressed shards into rank 0's HBM (OneStream: IPC window or RCCL).  The host arithmetic is covered on CPU by
tests/test_dist_gloo.py; checking a member against the oracle is the callers' business (tests/, bench.py) - nothing in
this package loads the oracle."""
import os
import struct


def shard_chunks(nchunks: int, world: int, rank: int):
    """contiguous [begin, end) chunk range of `rank` (sizes differ by at most one chunk)"""
    q, r = divmod(nchunks, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


GZIP_EXT_MAX = (1 << 32) - 1        # both size fields of a gzip-ext header are 32-bit (src/qatzip_gzip.c:98-118)


def member_plan(total_bytes: int, world: int, chunk: int, slice_bytes: int = 511 << 20):
    """BASELINE config 5 as a plan.  A logical buffer of 4 GiB or more is more than one gzip-ext member, so it becomes M
    members and is dealt to the ranks like a striped volume: member m is the logical range [m * world * slice,
    (m + 1) * world * slice), cut into `world` contiguous shards of whole chunks, shard r on rank r (the last member
    takes what is left: shard_chunks() spreads its chunks, the buffer's ragged end goes to the last rank that holds
    any).  All GPUs work on every member, a member's shards lie in rank order, and the members in order decode to the
    buffer in order - qzDecompress reads a sequence of members as it reads what qzCompress writes call after call.
    Returns per member [(logical offset, bytes)] by rank; world * slice stays below 4 GiB."""
    assert total_bytes >= 0 and world >= 1 and chunk >= 1
    cap = (GZIP_EXT_MAX // world) // chunk * chunk               # what a member's raw-size field allows per rank
    sl = max(chunk, min(slice_bytes, cap) // chunk * chunk)
    plan, base = [], 0
    while True:
        left = total_bytes - base
        if left >= world * sl:
            plan.append([(base + r * sl, sl) for r in range(world)])
            base += world * sl
            if base == total_bytes:
                break
            continue
        nch = (left + chunk - 1) // chunk                        # the last member: whole chunks per rank, the tail on the last
        shards = []
        for r in range(world):
            b, e = shard_chunks(nch, world, r)
            lo, hi = min(left, b * chunk), min(left, e * chunk)
            shards.append((base + lo, hi - lo))
        if left > 0 or not plan:
            plan.append(shards)
        break
    return plan


def local_offsets(plan, rank: int):
    """where rank `rank` keeps its shard of each member when its shards lie back to back in its own memory:
    [(local offset, bytes)] per member"""
    out, off = [], 0
    for shards in plan:
        n = shards[rank][1]
        out.append((off, n))
        off += n
    return out


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
        # the transport has a context of its own (its own streams): run_members() keeps a gather on the wire while the next
        # member is being deflated on ctx
        err = None
        self.tctx = tctx = None
        trace = os.environ.get("QATZIP_AMD_BENCH_TRACE")

        def step(what):
            if trace:
                import sys
                import time
                print("[OneStream %s, rank %d, t=%.1f] %s" % (transport, rank, time.time() % 1000, what), file=sys.stderr, flush=True)
        step("transport context")
        try:
            self.tctx = tctx = _lib.Context(ctx.device)
        except Exception as e:   # noqa: BLE001 - a rank that cannot get its context must not leave the others in the broadcast below
            err = "transport context: " + str(e)[:150]
        if not _all_ok(pg, err is None):
            self.error = err or "another rank could not make its transport context"
            self.close()
            return
        if transport == "rccl":
            idb = C.create_string_buffer(128)
            if rank == 0 and L.qzd_rccl_unique_id(idb) != 0:
                err = "ncclGetUniqueId failed (librccl.so.1 missing?)"
            uid = broadcast_bytes(pg, idb.raw if rank == 0 else None, 128, 0) if world > 1 else idb.raw
            if not _all_ok(pg, err is None):
                self.error = err or "rank 0 could not make an RCCL id"
                return
            if L.qzd_rccl_create(tctx.h, rank, world, uid, self.cap, C.byref(self.h)) != 0:
                err = "rccl init: " + L.qzd_last_error(tctx.h).decode()
        else:
            hbuf = C.create_string_buffer(64)
            step("window of %d MiB" % (self.cap >> 20))
            if rank == 0 and L.qzd_shard_root_create(tctx.h, world, self.cap, hbuf, C.byref(self.h)) != 0:
                err = "root window: " + L.qzd_last_error(tctx.h).decode()
            step("handle broadcast")
            handle = broadcast_bytes(pg, hbuf.raw if rank == 0 else None, 64, 0) if world > 1 else hbuf.raw
            step("attach")
            if rank != 0 and err is None and L.qzd_shard_attach(tctx.h, rank, world, handle, self.cap, C.byref(self.h)) != 0:
                err = "attach: " + L.qzd_last_error(tctx.h).decode()
            step("attached")
            if world > 1:
                # one slot per non-root rank (qzd_shard.hip: no allocation of the window is large - a window above 2 GiB never
                # came back from hipIpcOpenMemHandle); the root's handles travel as one blob, every rank maps its own
                blob = None
                if rank == 0:
                    blob = bytearray(64 * world)
                    if err is None:
                        for r in range(1, world):
                            hb = C.create_string_buffer(64)
                            if L.qzd_shard_slot_handle(self.h, r, hb) != 0:
                                err = "slot handle %d: %s" % (r, L.qzd_last_error(tctx.h).decode()); break
                            blob[64 * r:64 * r + 64] = hb.raw
                    blob = bytes(blob)
                blob = broadcast_bytes(pg, blob, 64 * world, 0)
                if rank != 0 and err is None and L.qzd_shard_attach_slot(self.h, blob[64 * rank:64 * rank + 64]) != 0:
                    err = "attach slot: " + L.qzd_last_error(tctx.h).decode()
                step("slot attached")
        if not _all_ok(pg, err is None):
            self.error = err or "another rank could not set the %s transport up" % transport
            self.close()
            return
        try:
            self.d_comp = ctx.alloc(_lib.max_deflate_len(shard_bytes, chunk))
        except Exception as e:   # noqa: BLE001
            err = "staging buffer: " + str(e)[:150]
        if not _all_ok(pg, err is None):
            self.error = err or "another rank could not allocate its staging buffer"
            self.close()

    def run(self, d_src, want_member=False):
        """one member.  Rank 0 gets {"raw_bytes", "member_bytes", "ms", "deflate_ms", "gather_ms", "crc32", ("stream")};
        the other ranks {"ms", ...}; {"error": ...} everywhere when any rank failed."""
        import ctypes as C
        import time
        import numpy as np
        L, ctx, rank, world, n, chunk = self.L, self.ctx, self.rank, self.world, self.n, self.chunk
        if self.error:
            return {"error": "this OneStream is unusable after an earlier failure: " + self.error}
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
        except Exception as e:   # noqa: BLE001
            err = "deflate: " + str(e)[:150]
        t1 = time.perf_counter()
        # agreement BEFORE the gather: a rank whose deflate failed must not leave the others waiting on the device
        if not _all_ok(self.pg, err is None):
            return {"error": err or "another rank's deflate failed"}
        try:
            dptr, slen, fcrc, raw = C.c_void_p(), C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
            if self.transport == "rccl":
                if L.qzd_rccl_gather(self.h, self.d_comp.ptr, clen, n, crc, self.level, C.byref(dptr), C.byref(slen),
                                     C.byref(fcrc), C.byref(raw)) != 0:
                    err = "gather: " + L.qzd_last_error(self.tctx.h).decode()
            else:
                if L.qzd_shard_put(self.h, self.d_comp.ptr, clen, n, crc, self.seq, self.timeout_s, None) != 0:
                    err = "put: " + L.qzd_last_error(self.tctx.h).decode()
                elif rank == 0 and L.qzd_shard_finish(self.h, self.seq, self.timeout_s, self.level, C.byref(dptr), C.byref(slen),
                                                      C.byref(fcrc), C.byref(raw)) != 0:
                    err = "finish: " + L.qzd_last_error(self.tctx.h).decode()
        except Exception as e:   # noqa: BLE001
            err = "gather: " + str(e)[:150]
        t2 = time.perf_counter()
        if not _all_ok(self.pg, err is None):
            self.error = err or "another rank failed"
            return {"error": self.error}
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

    def run_members(self, d_src, plan, d_out=None, depth=2):
        """The members of `plan` (member_plan(); this rank's shards lie back to back in d_src, local_offsets()), PIPELINED: while
        member m's shards are on the wire, member m + 1 is being deflated - the transport has a context (streams) of its
        own and a thread of its own, two staging buffers alternate.  The ranks agree after every deflate that all of them are
        going into the gather (a rank that failed never leaves the others waiting on the device); the transports' own
        waits are bounded.  Rank 0 appends every finished member to d_out (a DevBuf in its HBM) when given one.
        -> {"members", "raw_bytes", "out_bytes", "ms", "deflate_ms", "gather_ms", "overlapped_ms", "member_bytes": [...]}
        or {"error"}.  (overlapped_ms: time the gather thread was busy while the main thread was deflating.)"""
        import ctypes as C
        import queue
        import threading
        import time
        import numpy as np
        from . import _lib
        L, ctx, rank, world, chunk = self.L, self.ctx, self.rank, self.world, self.chunk
        if self.error:
            return {"error": "this OneStream is unusable after an earlier failure: " + self.error}
        M = len(plan)
        mine = local_offsets(plan, rank)
        nmax = max(n for _, n in mine)
        if nmax > self.n:
            return {"error": "a slice of the plan is larger than the shard this OneStream was made for"}
        bufs = [self.d_comp] + [ctx.alloc(_lib.max_deflate_len(self.n, chunk)) for _ in range(max(1, depth) - 1)]
        free = [threading.Semaphore(1) for _ in bufs]
        work = queue.Queue()
        gat = {"err": None, "ms": 0.0, "busy": [], "members": [], "out": 0, "raw": 0}

        def gather_thread():
            while True:
                item = work.get()
                if item is None:
                    return
                m, b, n, clen, crc, seq = item
                t0 = time.perf_counter()
                try:
                    if gat["err"] is None:
                        dptr, slen, fcrc, raw = C.c_void_p(), C.c_uint64(0), C.c_uint32(0), C.c_uint64(0)
                        if self.transport == "rccl":
                            if L.qzd_rccl_gather(self.h, bufs[b].ptr, clen, n, crc, self.level, C.byref(dptr), C.byref(slen),
                                                 C.byref(fcrc), C.byref(raw)) != 0:
                                gat["err"] = "member %d gather: %s" % (m, L.qzd_last_error(self.tctx.h).decode())
                        else:
                            if L.qzd_shard_put(self.h, bufs[b].ptr, clen, n, crc, seq, self.timeout_s, None) != 0:
                                gat["err"] = "member %d put: %s" % (m, L.qzd_last_error(self.tctx.h).decode())
                            elif rank == 0 and L.qzd_shard_finish(self.h, seq, self.timeout_s, self.level, C.byref(dptr),
                                                                  C.byref(slen), C.byref(fcrc), C.byref(raw)) != 0:
                                gat["err"] = "member %d finish: %s" % (m, L.qzd_last_error(self.tctx.h).decode())
                        if gat["err"] is None and rank == 0:
                            if d_out is not None:
                                if gat["out"] + slen.value > d_out.nbytes:
                                    gat["err"] = "member %d: the output buffer is full" % m
                                elif L.qzd_d2d(self.tctx.h, d_out.ptr + gat["out"], dptr.value, slen.value) != 0:
                                    gat["err"] = "member %d: copy out of the window failed" % m
                            gat["members"].append(slen.value); gat["out"] += slen.value; gat["raw"] += raw.value
                except Exception as e:   # noqa: BLE001
                    gat["err"] = "member %d: %s" % (m, str(e)[:120])
                finally:
                    t1 = time.perf_counter()
                    gat["ms"] += (t1 - t0) * 1e3; gat["busy"].append((t0, t1))
                    free[b].release()

        th = threading.Thread(target=gather_thread, daemon=True)
        th.start()
        if self.pg is not None:
            self.pg.barrier()
        t_begin = time.perf_counter()
        t_defl, defl_spans, err = 0.0, [], None
        for m, (off, n) in enumerate(mine):
            b = m % len(bufs)
            free[b].acquire()                                   # the gather of member m - depth has let go of this buffer
            t0 = time.perf_counter()
            clen = crc = 0
            try:
                v = _lib.DevBuf.__new__(_lib.DevBuf); v.ctx, v.nbytes, v.ptr = ctx, n, d_src.ptr + off
                # the member's stream closes with the shard of the last rank that holds any of it (rank 0 for an empty member)
                closer = max([r for r in range(world) if plan[m][r][1]] or [0])
                if n or rank == closer:
                    clen, crcs = ctx.deflate_raw(v, n, chunk, self.level, 1 if rank == closer else 0, bufs[b])
                    crcs = np.ascontiguousarray(crcs, dtype=np.uint32)
                    crc = L.qzd_crc32_fold(crcs.ctypes.data, len(crcs), chunk, n)
            except Exception as e:   # noqa: BLE001
                err = "member %d deflate: %s" % (m, str(e)[:150])
            t1 = time.perf_counter()
            t_defl += (t1 - t0) * 1e3; defl_spans.append((t0, t1))
            # every rank learns here whether every rank is going into the gather of this member (nobody waits on the device
            # for a rank that failed); a transport that failed on an earlier member ends the run the same way
            if not _all_ok(self.pg, err is None and gat["err"] is None):
                free[b].release()
                err = err or gat["err"] or "another rank failed before member %d" % m
                break
            # the member's sequence number is the same on every rank whatever happens to the member: counted here, per
            # member queued, not by the gather thread on success (after a failure the ranks' numbers diverged)
            self.seq += 1
            work.put((m, b, n, clen, crc, self.seq))
        work.put(None)
        th.join()
        t_end = time.perf_counter()
        ok = _all_ok(self.pg, err is None and gat["err"] is None)
        for extra in bufs[1:]:
            extra.free()
        if not ok:
            # a gather that failed on one rank may have been entered by the others (the agreement before member m cannot know
            # how member m - 1's gather, still in flight, will end): their transport waits are bounded, and nobody uses this
            # stream's window / communicator again
            self.error = err or gat["err"] or "another rank failed"
            return {"error": self.error}
        if self.pg is not None:
            self.pg.barrier()
        overl = sum(max(0.0, min(g1, d1) - max(g0, d0)) for g0, g1 in gat["busy"] for d0, d1 in defl_spans) * 1e3
        out = {"members": M, "ms": (t_end - t_begin) * 1e3, "deflate_ms": t_defl, "gather_ms": gat["ms"], "overlapped_ms": overl}
        if rank == 0:
            out.update({"raw_bytes": gat["raw"], "out_bytes": gat["out"], "member_bytes": gat["members"]})
        return out

    def close(self):
        if self.h:
            (self.L.qzd_rccl_close if self.transport == "rccl" else self.L.qzd_shard_close)(self.h)
            self.h = None
        if getattr(self, "d_comp", None) is not None:
            self.d_comp.free(); self.d_comp = None
        if getattr(self, "tctx", None) is not None and self.tctx is not self.ctx:
            self.tctx.close(); self.tctx = None


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


def allgather_floats(pg, vals):
    """every rank's list of floats, rank by rank (a world of one: its own)"""
    if pg is None:
        return [list(vals)]
    import torch
    world = pg.get_world_size()
    mine = torch.tensor(list(vals), dtype=torch.float64)
    out = [torch.zeros_like(mine) for _ in range(world)]
    pg.all_gather(out, mine)
    return [[float(x) for x in t] for t in out]
