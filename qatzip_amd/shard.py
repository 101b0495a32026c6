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


def allreduce(pg, v: float, op: str) -> float:
    if pg is None:
        return v
    import torch
    t = torch.tensor([v], dtype=torch.float64)
    pg.all_reduce(t, op=getattr(pg.ReduceOp, op))
    return float(t[0])
