"""N>1 path on CPU: world_size-2 gloo group exercising the shard arithmetic, the 16 B/rank record exchange and the
CRC / offset fold that turn per-rank compressed shards into ONE gzip-ext stream.  The per-rank codec
here is the oracle (no GPU in this container); on the GPU box the same host logic sits on top of qzd_deflate_raw."""
import os
import struct
import sys
import zlib

import torch.multiprocessing as mp

import datagen
import oracle_lib as O
from qatzip_amd import shard as S

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 8, 65, 32768):
        for w in (1, 2, 3, 8):
            rs = [S.shard_chunks(n, w, r) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(e - b for b, e in rs) - min(e - b for b, e in rs) <= 1


def test_crc32_combine_matches_zlib():
    a, b = datagen.gen_bytes("text", 70001, 1), datagen.gen_bytes("rand", 12345, 2)
    assert S.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert S.crc32_combine(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hw = 16384
    src = datagen.gen_bytes("silesia", 10 * hw + 777, 5)
    nchunks = (len(src) + hw - 1) // hw
    b, e = S.shard_chunks(nchunks, world, rank)
    mine = src[b * hw:min(e * hw, len(src))]
    last = 1 if rank == world - 1 else 0
    rc, used, comp, _ = O.sw_compress("RAW", mine, hw, 1, last=last)      # stand-in for qzd_deflate_raw on this rank's GPU
    assert rc == 0 and used == len(mine)
    recs = S.all_gather_records(dist, S.pack_record(len(mine), len(comp), zlib.crc32(mine)), world)
    offs, raw, total, crc = S.fold_records(recs)
    t = S.allreduce(dist, float(rank + 1), "MAX")
    # what bench.py's top level reports for N > 1: every rank's own pass times, rank by rank
    assert S.allgather_floats(dist, [rank + 0.5, 10.0 * rank]) == [[r + 0.5, 10.0 * r] for r in range(world)]
    q.put((rank, offs[rank], comp, raw, total, crc, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_build_one_stream():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    src = datagen.gen_bytes("silesia", 10 * 16384 + 777, 5)
    _, _, _, raw, total, crc, tmax = got[0]
    stream = bytearray(total)
    for rank, off, comp, *_ in got:
        stream[off:off + len(comp)] = comp
    assert raw == len(src) and tmax == 2.0
    assert crc == (zlib.crc32(src) & 0xffffffff)
    # the sharded stream is the single-stream software-path output, byte for byte
    assert bytes(stream) == O.sw_compress("RAW", src, 16384, 1)[2]
    gz = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 4, 255, 12, 0]) + b"QZ\x08\x00" + struct.pack("<II", raw, total) + \
        bytes(stream) + struct.pack("<II", crc, raw)
    assert gz == O.sw_compress("GZIP_EXT", src, 16384, 1)[2]


# ---- config 5 as written: a buffer of more than one member, M members of `world` shards each (shard.member_plan) ----
def test_member_plan_covers_the_buffer_in_order():
    for total, world, chunk, sl in ((8 * (8176 << 20), 8, 65536, 511 << 20), (3 * (5 * 65536) + 17, 3, 65536, 2 * 65536), (0, 2, 4096, 8192),
                                    (1 << 20, 1, 16384, 1 << 18), (8 * (1 << 32) + 12345, 8, 65536, 1 << 30), (70000, 4, 65536, 65536)):
        plan = S.member_plan(total, world, chunk, sl)
        flat = [x for m in plan for x in m]
        assert all(len(m) == world for m in plan)
        assert sum(n for _, n in flat) == total
        assert [o for o, _ in flat] == [sum(n for _, n in flat[:i]) for i in range(len(flat))]          # the logical buffer, in order
        assert all(o % chunk == 0 for o, n in flat if n) and all(sum(n for _, n in m) <= 0xffffffff for m in plan)
        for m in plan:                                                                                  # whole chunks, but for the member's last bytes
            nz = [n for _, n in m if n]
            assert all(n % chunk == 0 for n in nz[:-1])
        for r in range(world):
            loc = S.local_offsets(plan, r)
            assert [o for o, _ in loc] == [sum(n for _, n in loc[:i]) for i in range(len(loc))]
    assert len(S.member_plan(8 * (8176 << 20), 8, 65536)) == 16                                         # BASELINE config 5: 16 members


HW_M, SL_M, TOTAL_M = 16384, 3 * 16384, 2 * (7 * 16384) + 333        # three members over two ranks: 6 + 6 + (2 chunks and a bit)


def _member_worker(rank, world, port, q, total=TOTAL_M, hw=HW_M, sl=SL_M):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = S.member_plan(total, world, hw, sl)
    logical = datagen.gen_bytes("silesia", total, 9)
    members = []
    for m, shards in enumerate(plan):
        off, n = shards[rank]
        mine = logical[off:off + n]                                    # the striped volume: my shard of member m
        last_holder = max(r for r in range(world) if shards[r][1] or r == 0)
        rc, used, comp, _ = O.sw_compress("RAW", mine, hw, 1, last=1 if rank == last_holder else 0) if (n or rank == last_holder) else (0, 0, b"", 0)
        assert rc == 0 and used == len(mine)
        recs = S.all_gather_records(dist, S.pack_record(len(mine), len(comp), zlib.crc32(mine)), world)
        offs, raw, total, crc = S.fold_records(recs)
        members.append((offs[rank], comp, raw, total, crc))
    q.put((rank, members))
    dist.barrier()
    dist.destroy_process_group()


def _members_against_the_software_path(world, total, hw, sl, nmembers, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_member_worker, args=(r, world, port, q, total, hw, sl)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    logical = datagen.gen_bytes("silesia", total, 9)
    plan = S.member_plan(total, world, hw, sl)
    assert len(plan) == nmembers
    HW_M = hw
    out, expect, pos = b"", b"", 0
    for m, shards in enumerate(plan):
        _, _, raw, total, crc = got[0][m]
        stream = bytearray(total)
        for r in range(world):
            off, comp, *_ = got[r][m]
            stream[off:off + len(comp)] = comp
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 4, 255, 12, 0]) + b"QZ\x08\x00" + struct.pack("<II", raw, total) + \
            bytes(stream) + struct.pack("<II", crc, raw)
        n_m = sum(n for _, n in shards)
        expect += O.sw_compress("GZIP_EXT", logical[pos:pos + n_m], HW_M, 1)[2]    # one qzCompress call per member, in order
        pos += n_m
    assert pos == len(logical) and out == expect
    rc, used, back = O.sw_decompress("GZIP_EXT", out, len(logical) + 64)            # and the sequence decodes to the buffer, in order
    assert rc == 0 and used == len(out) and back == logical


def test_three_members_from_two_ranks_are_what_the_software_path_writes():
    _members_against_the_software_path(2, TOTAL_M, HW_M, SL_M, 3, 31500 + os.getpid() % 2000)


def test_eight_ranks_ragged_members():
    """BASELINE config 5's shape in small: EIGHT ranks, members whose last one is ragged - three ranks hold a chunk of it, one
    the last 123 bytes, four nothing at all (they still take part in the record exchange, the rank before them closes
    the member's stream) - against one qzCompress call per member of the software path (src/qatzip.c:1691-1718: the
    engine's in-order retire across accelerators)"""
    hw, sl = 4096, 2 * 4096
    total = 2 * (8 * sl) + 3 * hw + 123                                             # two full members, then three chunks and 123 bytes
    plan = S.member_plan(total, 8, hw, sl)
    tail = [n for _, n in plan[-1]]
    assert sum(tail) == 3 * hw + 123 and tail.count(0) == 4 and sorted(x for x in tail if x) == [123, hw, hw, hw]
    for r in range(8):                                                              # every rank's own bytes lie back to back in its buffer
        loc = S.local_offsets(plan, r)
        assert [o for o, _ in loc] == [sum(n for _, n in loc[:i]) for i in range(len(loc))]
        assert [n for _, n in loc] == [plan[m][r][1] for m in range(len(plan))]
    _members_against_the_software_path(8, total, hw, sl, 3, 33500 + os.getpid() % 2000)
