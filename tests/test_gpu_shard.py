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
