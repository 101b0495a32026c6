#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X.

Workload (BASELINE.json configs[1]): QZ_DEFLATE_GZIP_EXT, level 1, hw_buff_sz 64 KB, a 4 GB synthetic
"Silesia-like" buffer per GPU, handled as 2 calls of 2 GiB (qatzip lengths are 32-bit), inputs already
resident in HBM when the timed region starts.  One step = one pass of the hot path over that buffer:
compress every call, then decompress every call (the reference harness' "-D both", test/main.c:2204-2299).
`value` = uncompressed bytes moved in both directions by all ranks / max-over-ranks wall time (the reference
counts uncompressed bytes for either direction and doubles them for "both", test/main.c:2336-2346).
Multi-GPU: independent chunks shard across ranks with no data-path collective (weak scaling: every rank
owns its own buffer); gloo carries the barrier and the max/sum reductions of the timings.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

CHUNK = 65536
CALL_BYTES = 1 << 31            # one qzCompress-sized call (2 GiB)
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def dist_setup():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist
    return rank, world, local, pg


def barrier(pg):
    if pg is not None:
        pg.barrier()


def allreduce(pg, v, op):
    from qatzip_amd import shard
    return shard.allreduce(pg, v, op)


def cpu_baseline(sample: bytes):
    """The oracle (a port of the reference's SW path) timed on this box's host cores, 1 thread, bounded sample."""
    import oracle_lib as O
    t0 = time.perf_counter()
    rc, used, out, _ = O.sw_compress("GZIP_EXT", sample, CHUNK, 1, cap=len(sample) * 9 // 8 + 65536)
    t1 = time.perf_counter()
    assert rc == 0 and used == len(sample)
    rc, cused, back = O.sw_decompress("GZIP_EXT", out, len(sample) + 64)
    t2 = time.perf_counter()
    assert rc == 0 and back == sample
    both = 2 * len(sample) / (t2 - t0) / 1e9
    return {"value": round(both, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "compress": round(len(sample) / (t1 - t0) / 1e9, 4), "decompress": round(len(sample) / (t2 - t1) / 1e9, 4),
            "sample": "%d MiB of the same buffer, GZIP_EXT L1 64 KB chunks, compress + decompress, "
                      "oracle/libqzoracle.so, 1 thread" % (len(sample) >> 20),
            "ratio": round(len(out) / len(sample), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mb", type=int, default=4096, help="buffer size per GPU in MiB (default: the 4 GB config)")
    ap.add_argument("--base-mb", type=int, default=128, help="distinct synthetic data generated per GPU (tiled)")
    ap.add_argument("--cpu-mb", type=int, default=8, help="sample size for the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank, world, local, pg = dist_setup()
    import datagen
    import qatzip_amd

    # one process per GPU: LOCAL_RANK picks the device (QATZIP_AMD_BENCH_DEVICE overrides it, for exercising the
    # multi-rank path on a box with fewer GPUs than ranks)
    ctx = qatzip_amd.Context(int(os.environ.get("QATZIP_AMD_BENCH_DEVICE", local)))
    total = args.mb << 20
    base_n = min(args.base_mb << 20, total)
    base = datagen.gen("silesia", base_n, 20250523 + rank)
    d_src = ctx.alloc(total)
    for off in range(0, total, base_n):       # tile: chunks are independent, nothing is shared across tiles
        d_src.upload(base[:min(base_n, total - off)], off)
    ncalls = (total + CALL_BYTES - 1) // CALL_BYTES
    call_n = [min(CALL_BYTES, total - i * CALL_BYTES) for i in range(ncalls)]
    d_comp = [ctx.alloc(qatzip_amd.max_deflate_len(n, CHUNK)) for n in call_n]
    d_back = ctx.alloc(max(call_n))

    def view(buf, off, n):
        v = qatzip_amd.DevBuf.__new__(qatzip_amd.DevBuf)
        v.ctx, v.nbytes, v.ptr = buf.ctx, n, buf.ptr + off
        return v

    comp_len = [0] * ncalls

    def compress_all():
        for i, n in enumerate(call_n):
            ctx.deflate_raw_async(view(d_src, i * CALL_BYTES, n), n, CHUNK, 1, 1, d_comp[i])
            ctx.sync()
            comp_len[i] = ctx.result()

    def decompress_all():
        for i, n in enumerate(call_n):
            iu, ol, crc = ctx.inflate_stream(d_comp[i], comp_len[i], d_back, CHUNK, want_crc=True)
            assert iu == comp_len[i] and ol == n

    def step():
        compress_all()
        decompress_all()

    for _ in range(args.warmup):
        step()
    # parity guard outside the timed region: what came back is what went in
    assert ctx.crc32(d_back, call_n[-1]) == ctx.crc32(view(d_src, (ncalls - 1) * CALL_BYTES, call_n[-1]), call_n[-1])

    ctx.k1_stats(reset=True)
    barrier(pg); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync(); barrier(pg)
    dt = allreduce(pg, time.perf_counter() - t0, "MAX")
    # dominant kernel (K1, the LZ77 parse): HIP events around every launch of the timed region, on its own stream
    k1_ms, k1_launches, k1_chunks = ctx.k1_stats()

    # per-direction split (untimed by the contract, reported alongside) and one K1 launch alone on the chip
    ctx.sync(); t1 = time.perf_counter(); compress_all(); ctx.sync(); tc = time.perf_counter() - t1
    t1 = time.perf_counter(); decompress_all(); ctx.sync(); td = time.perf_counter() - t1
    inf_ms = ctx.inflate_timing()
    batch_chunks = ctx.batch_chunks()         # chunks per full K1 launch (three rounds over the resident workgroups)
    probe_n = min(batch_chunks * CHUNK, call_n[0])
    ctx.deflate_raw_async(view(d_src, 0, probe_n), probe_n, CHUNK, 1, 1, d_comp[0]); ctx.sync()
    k_ms = ctx.timing()                      # single batch => K1 ran alone on the chip
    comp_total = allreduce(pg, float(sum(comp_len)), "SUM")
    raw_total = float(total) * world
    tc = allreduce(pg, tc, "MAX"); td = allreduce(pg, td, "MAX")

    if rank == 0:
        # HBM traffic per K1 launch: PMC counters cannot be read from inside this process; they come from the committed
        # rocprofv3 --pmc passes of this same command (profiles/r1_pmc.json, tools/pmc_summary.py): FETCH_SIZE and
        # WRITE_SIZE collected in separate runs, KiB -> bytes, FETCH x2 per the gfx950 note; quoted only when the
        # profiled command had the same launch mix (same --mb).
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc.json")) as f:
                pj = json.load(f)
            pk = pj["kernels"].get(pj.get("k1_key", ""))
            if pk and pj.get("bench_mb") == args.mb and pj.get("k1_launch_chunks") == batch_chunks:
                traffic = pk["hbm_bytes_fetch_x2"]
        except (OSError, KeyError, ValueError):
            pass
        value = 2.0 * raw_total * args.steps / dt / 1e9
        ratio = comp_total / raw_total
        # algorithmic bytes of the K1 launches of the timed region: every input byte read once, every compressed byte
        # written once (U + C per chunk), divided over the launches; rank 0's own launches and time
        alg_total = args.steps * (float(total) + float(sum(comp_len)))
        alg_bytes = alg_total / max(k1_launches, 1)
        launch_ms = k1_ms / max(k1_launches, 1)
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        res = {
            "metric": "compress + decompress GB/s (input bytes), QZ_DEFLATE_GZIP_EXT L1, 64 KB chunks",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "QZ_DEFLATE_GZIP_EXT level 1, 64 KB chunks, %d MiB Silesia-like buffer per GPU "
                                   "(%d MiB distinct, tiled), %d call(s) of <= 2 GiB, compress then decompress" %
                                   (args.mb, base_n >> 20, ncalls),
                       "chunk": CHUNK, "ratio": round(ratio, 4), "parallelism": "chunks sharded over %d rank(s), "
                       "no data-path collective" % world,
                       "compress_GBps": round(raw_total / tc / 1e9, 3), "decompress_GBps": round(raw_total / td / 1e9, 3)},
            "roofline": {"bound": "hbm", "kernel": "qzk_lz77_pull_kernel", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "algorithmic_bytes": int(alg_bytes),
                         "launch_ms": round(launch_ms, 3), "launches": int(k1_launches),
                         "chunks_per_launch": round(k1_chunks / max(k1_launches, 1), 1),
                         "full_launch_alone_ms": round(k_ms[0], 3), "full_launch_chunks": probe_n // CHUNK,
                         "other_kernels_ms": {"qzk_huff_kernel": round(k_ms[1], 3), "scan+gather": round(k_ms[2], 3),
                                              "qzk_inflate_tok_kernel+qzk_lz_resolve_kernel(last call)": round(inf_ms[0], 3),
                                              "of which qzk_lz_resolve_kernel": round(inf_ms[2], 3),
                                              "qzk_crc_kernel(last call)": round(inf_ms[1], 3)}},
        }
        if not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline(base[:args.cpu_mb << 20].tobytes())
        print(json.dumps(res))
    if pg is not None:
        pg.destroy_process_group()


if __name__ == "__main__":
    main()
