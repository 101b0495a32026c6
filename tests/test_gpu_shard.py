"""ONE gzip-ext member from several ranks (BASELINE config 5's shape): every rank deflates its shard on its
GPU, the compressed shards travel to rank 0's HBM - as peer-to-peer copies into an IPC window (qzd_shard_*) or as one
RCCL send/recv group (qzd_rccl_*) - rank 0 folds the CRCs and closes the member, which must be, byte for byte, what one
software-path qzCompress over the whole buffer writes (src/qatzip_sw.c:77-256; the in-order retire it stands for:
src/qatzip.c:1691-1718).  One process per rank, gloo for the handle / id.  With fewer GPUs than ranks the ranks of the
IPC test share a device (the IPC path is the same); the tests that need real peers skip on a one-GPU box."""
import os
import sys

import pytest
import torch.multiprocessing as mp

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _ndev():
    import qatzip_amd
    return qatzip_amd.load().qzd_device_count()


def _worker(rank, world, port, n, chunk, level, transport, distinct, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qatzip_amd
    from qatzip_amd import shard as S
    ndev = qatzip_amd.load().qzd_device_count()
    assert not distinct or ndev >= world
    ctx = qatzip_amd.Context(rank % ndev)
    whole = datagen.gen("silesia", world * n, 77)
    d_src = ctx.alloc(n); d_src.upload(whole[rank * n:(rank + 1) * n])
    one = S.OneStream(ctx, dist, rank, world, n, chunk, level, transport)
    if one.error:
        if rank == 0:
            q.put(("error", one.error))
    else:
        for _ in range(2):                                   # twice: the second member reuses nothing stale
            res = one.run(d_src, want_member=True)
            if rank == 0:
                q.put(("ok", res.get("stream"), res.get("raw_bytes"), ndev, res.get("error")))
        one.close()
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def _run(world, level, transport, distinct, n=5 * 65536, chunk=65536):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + 7 * world + 13 * level + (3 if transport == "rccl" else 0)) % 2000
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, chunk, level, transport, distinct, q)) for r in range(world)]
    for p in ps:
        p.start()
    first = q.get(timeout=300)
    got = [first] if first[0] == "error" else [first, q.get(timeout=300)]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    whole = datagen.gen_bytes("silesia", world * n, 77)
    exp = O.sw_compress("GZIP_EXT", whole, chunk, level, cap=len(whole) * 9 // 8 + 65536)[2]
    return got, whole, exp


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_build_one_member_over_peer_copies(world):
    got, whole, exp = _run(world, 1, "ipc", False)
    for tag, stream, raw, ndev, err in got:
        assert tag == "ok" and not err and raw == len(whole)
        assert stream == exp, (len(stream), len(exp))
    print("one member from %d ranks on %d device(s): %d -> %d bytes, identical to the software path's" % (world, got[0][3], len(whole), len(exp)))


def test_member_header_follows_the_level():
    """advisor (round 2): the member a multi-rank job closes carries the XFL byte of its level, as qzCompress writes it"""
    got, whole, exp = _run(2, 6, "ipc", False, n=3 * 65536)
    for tag, stream, raw, ndev, err in got:
        assert tag == "ok" and not err and stream == exp and stream[8] == 0


@pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs: hipIpcOpenMemHandle(LazyEnablePeerAccess) across real peers")
def test_peer_copies_between_two_gpus():
    got, whole, exp = _run(2, 1, "ipc", True)
    for tag, stream, raw, ndev, err in got:
        assert tag == "ok" and not err and stream == exp


def test_rccl_transport_single_rank():
    """the RCCL entry points on whatever this box has: a world of one (ncclCommInitRank, the all-gather of the record, the
    member closed by the root) - bytes equal to the software path's"""
    got, whole, exp = _run(1, 1, "rccl", False)
    assert got[0][0] == "ok", got[0]
    for tag, stream, raw, ndev, err in got:
        assert not err and stream == exp


@pytest.mark.skipif(_ndev() < 2, reason="RCCL refuses two ranks on one device")
def test_rccl_transport_between_gpus():
    world = min(_ndev(), 4)
    got, whole, exp = _run(world, 1, "rccl", True)
    for tag, stream, raw, ndev, err in got:
        assert tag == "ok" and not err and stream == exp


# ---- BASELINE config 5 as written: a buffer of several members, every member built by all ranks, member m + 1 deflated while
# member m's shards are on the wire (OneStream.run_members) ----
CH_M, SLICE_M = 65536, 3 * 65536


def _members_worker(rank, world, port, total, transport, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qatzip_amd
    from qatzip_amd import shard as S
    ndev = qatzip_amd.load().qzd_device_count()
    ctx = qatzip_amd.Context(rank % ndev)
    logical = datagen.gen("silesia", total, 78)
    plan = S.member_plan(total, world, CH_M, SLICE_M)
    mine = np.concatenate([logical[o:o + n] for o, n in (m[rank] for m in plan)] + [np.zeros(0, np.uint8)])   # my shards, back to back
    d_src = ctx.alloc(max(1, mine.size)); d_src.upload(mine)
    d_out = ctx.alloc(total * 9 // 8 + 4096 * len(plan)) if rank == 0 else None
    one = S.OneStream(ctx, dist, rank, world, max(n for m in plan for _, n in m), CH_M, 1, transport)
    if one.error:
        if rank == 0:
            q.put(("error", one.error))
    else:
        res = one.run_members(d_src, plan, d_out)
        if rank == 0:
            q.put(("ok", d_out.download(res["out_bytes"]).tobytes() if "error" not in res else None, res))
        one.close()
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world,total", [(3, 3 * (3 * SLICE_M) + 2 * CH_M + 4321), (2, 4 * (2 * SLICE_M)),
                                         (8, 2 * (8 * SLICE_M) + 3 * CH_M + 321)])       # eight ranks, a ragged third member: four ranks hold nothing of it
def test_members_of_a_larger_buffer_follow_each_other_like_calls(world, total):
    """>= 3 members x 3 ranks (sharing this box's GPU: the IPC window is the same path): the concatenation is, byte for
    byte, what the software path writes for one qzCompress call per member (src/qatzip_sw.c:77-256, the in-order retire it
    stands for: src/qatzip.c:1691-1718), and qzDecompress reads the sequence back to the buffer"""
    import qatzip_amd.api as A
    from qatzip_amd import shard as S
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33600 + (os.getpid() + 11 * world) % 2000
    ps = [ctx.Process(target=_members_worker, args=(r, world, port, total, "ipc", q)) for r in range(world)]
    for p in ps:
        p.start()
    tag, stream, res = q.get(timeout=300)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert tag == "ok" and "error" not in res, res
    logical = datagen.gen_bytes("silesia", total, 78)
    plan = S.member_plan(total, world, CH_M, SLICE_M)
    assert res["members"] == len(plan) >= 3 and res["raw_bytes"] == total
    exp, pos = b"", 0
    for m in plan:
        n_m = sum(n for _, n in m)
        exp += O.sw_compress("GZIP_EXT", logical[pos:pos + n_m], CH_M, 1, cap=n_m * 9 // 8 + 65536)[2]
        pos += n_m
    assert stream == exp, (len(stream), len(exp))
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, CH_M)
    rc, used, back = s.decompress(stream, total + 64)
    s.close()
    assert rc == 0 and used == len(stream) and back == logical
    print("%d members from %d ranks: %d -> %d bytes; deflate %.1f ms, gathers %.1f ms of which %.1f ms beside a deflate" %
          (res["members"], world, total, len(stream), res["deflate_ms"], res["gather_ms"], res["overlapped_ms"]))
