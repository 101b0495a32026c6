"""ONE gzip-ext member from several ranks (BASELINE config 5's shape): every rank deflates its shard on its
GPU, the compressed shards travel to rank 0's HBM as peer-to-peer copies into an IPC window (qzd_shard_*), rank 0 folds
the CRCs and closes the member - which must be, byte for byte, what one software-path qzCompress over the whole buffer
writes (src/qatzip_sw.c:77-256; the in-order retire it stands for: src/qatzip.c:1691-1718).  One process per rank, gloo
for the 64-byte handle; with fewer GPUs than ranks the ranks share a device (the IPC path is the same)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

import datagen
import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, n, chunk, q):
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qatzip_amd
    from qatzip_amd import shard as S
    ndev = qatzip_amd.load().qzd_device_count()
    ctx = qatzip_amd.Context(rank % ndev)
    whole = datagen.gen("silesia", world * n, 77)
    d_src = ctx.alloc(n); d_src.upload(whole[rank * n:(rank + 1) * n])
    for seq in (1, 2):                                   # twice: the second stream reuses nothing stale
        res = S.one_stream(ctx, dist, rank, world, d_src, n, chunk, verify="full", seq=seq)
        if rank == 0:
            q.put((seq, res["verified"], res["stream"], res["raw_bytes"], ndev))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_build_one_member_over_peer_copies(world):
    n, chunk = 5 * 65536, 65536
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + world) % 2000
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, chunk, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    whole = datagen.gen_bytes("silesia", world * n, 77)
    exp = O.sw_compress("GZIP_EXT", whole, chunk, 1, cap=len(whole) * 9 // 8 + 65536)[2]
    for seq, verified, stream, raw, ndev in got:
        assert verified and raw == len(whole)
        assert stream == exp, (seq, len(stream), len(exp))
    print("one member from %d ranks on %d device(s): %d -> %d bytes, identical to the software path's" % (world, got[0][4], len(whole), len(exp)))
