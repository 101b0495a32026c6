#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X.

Workload (BASELINE.json configs[1]): QZ_DEFLATE_GZIP_EXT, level 1, hw_buff_sz 64 KB, a 4 GB synthetic
"Silesia-like" buffer per GPU, handled as ONE device-layer call of 4 GiB per direction (65536 chunks in one launch; the
device ABI's lengths are 64-bit - through qatzip.h, whose lengths are 32-bit, the same buffer is two calls, see
config.api_* and config.concurrent_sessions), inputs already resident in HBM when the timed region starts.  One step = one pass of the hot path over that buffer:
compress every call, then decompress every call (the reference harness' "-D both", test/main.c:2204-2299).
`value` = uncompressed bytes moved in both directions by all ranks / max-over-ranks wall time (the reference
counts uncompressed bytes for either direction and doubles them for "both", test/main.c:2336-2346).
Multi-GPU: independent chunks shard across ranks with no data-path collective in the timed region (weak scaling:
every rank owns its own buffer); gloo carries the barrier and the max/sum reductions of the timings.  After the timed
region the ranks also build gzip-ext members the way BASELINE config 5 asks for - a striped volume: member m holds one
slice of every rank (--members, default 16 from 8 ranks on, else 4), the members are PIPELINED (while member m travels to
rank 0's HBM the ranks compress member m + 1) - over peer-to-peer copies into an IPC window and over an RCCL send/recv
group, both timed and both verified by decoding on rank 0: `config.one_stream`.  `python bench.py --gpus N` launches its own
N ranks when no launcher set WORLD_SIZE.

Beside the headline the line carries (rank 0, N = 1 only, all outside the timed region):
  config.api_*          the same work through qatzip.h itself: ONE qzCompress / qzDecompress call of --api-mb (2047) MiB on
                        qzMalloc(PINNED_MEM) buffers, PCIe both ways included, and its ratio to min(link, kernel rate)
                        (the decode takes the member in growing pieces while it arrives: profiles/r5_api_decompress_pieces.txt)
  config.pcie_*         plain pinned hipMemcpyAsync, 1 GiB each way, on this box
  config.concurrent_sessions   the buffer as 2 GiB calls of two sessions started together (the harness' -t)
  config.raw_sweep      BASELINE config 3: QZ_DEFLATE_RAW, hw_buff_sz 16 / 64 / 128 KB - call rates, the kernels' own
                        milliseconds (HIP events inside the library), their fraction of the HBM peak and their measured HBM
                        traffic per launch (profiles/r6_raw<K>_pmc.json)
  config.lz4            BASELINE config 4: LZ4 frames of 64 KB with XXH32, the same way (profiles/r6_lz4_pmc.json)
  roofline              the dominant kernel (K1, qzk_lz77_pull_kernel) against the HBM peak (datasheet) and the copy rate
                        measured on this box; traffic = HBM bytes per launch from profiles/r6_pmc.json
  roofline_decode       the two inflate kernels of a call (phase A + phase B), the same way
  cpu_baseline          the software path's port on every physical core of this host, and this host's own libz
`traffic` fields are read from the committed rocprofv3 --pmc summaries and only when those were taken from THESE sources
(every summary carries the SHA-256 of qatzip_amd/csrc; src_sha256() below) with this command's --mb - null otherwise.

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
CALL_BYTES = 1 << 32            # one device-layer call: the device ABI's lengths are 64-bit (deflate: at most 4 GiB a call)
API_CALL_BYTES = 1 << 31        # the same job through qatzip.h, whose lengths are 32-bit: calls of 2 GiB
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
TILE_SKEW = 4099                # the distinct data is tiled with a period that is no multiple of any chunk size


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


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: become the launcher - one process per GPU (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, exactly what torch.distributed.run sets), gloo rendezvous on
    127.0.0.1, rank 0's JSON line passed through.  Under torchrun (WORLD_SIZE already set) this is never reached."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                   stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in ps:
        rc = p.wait() or rc
    sys.exit(rc)


def barrier(pg):
    if pg is not None:
        pg.barrier()


def allreduce(pg, v, op):
    from qatzip_amd import shard
    return shard.allreduce(pg, v, op)


# ------------------------------------------------------------------ CPU baseline (reported, not the target)
def host_cpu():
    """(one logical CPU per physical core this process may use, model name)"""
    model, cores = "unknown", {}
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    try:
        proc = phys = core = None
        for line in list(open("/proc/cpuinfo")) + [""]:
            k, _, v = line.partition(":")
            k = k.strip(); v = v.strip()
            if k == "model name":
                model = v
            elif k == "processor":
                proc = int(v)
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif k == "" and proc is not None:
                if proc in allowed:
                    key = (phys, core) if phys is not None and core is not None else ("p", proc)
                    cores[key] = min(cores.get(key, proc), proc)
                proc = phys = core = None
    except (OSError, ValueError):
        pass
    cpus = sorted(cores.values()) or sorted(allowed)
    return cpus, model


class _ZStream(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("next_in", _C.c_void_p), ("avail_in", _C.c_uint), ("total_in", _C.c_ulong), ("next_out", _C.c_void_p),
                ("avail_out", _C.c_uint), ("total_out", _C.c_ulong), ("msg", _C.c_char_p), ("state", _C.c_void_p),
                ("zalloc", _C.c_void_p), ("zfree", _C.c_void_p), ("opaque", _C.c_void_p), ("data_type", _C.c_int),
                ("adler", _C.c_ulong), ("reserved", _C.c_ulong)]


def _libz():
    """this host's libz through its C API (no CPython buffer in between), or None"""
    import ctypes as C
    try:
        lib = C.CDLL("libz.so.1")
    except OSError:
        return None
    lib.zlibVersion.restype = C.c_char_p
    lib.deflateInit2_.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.inflateInit2_.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    for f in (lib.deflate, lib.inflate):
        f.argtypes = [C.c_void_p, C.c_int]
    for f in (lib.deflateEnd, lib.inflateEnd):
        f.argtypes = [C.c_void_p]
    return lib


def _libz_pass(lib, src_buf, n, dst_buf, dst_cap, back_buf):
    """the software path's loop (src/qatzip_sw.c:178-231, :323-351) on libz itself: deflateInit2(1, 8, -15, 9, 0), one
    deflate(Z_FULL_FLUSH) per 64 KB with the whole destination behind it, Z_FINISH on the last; then one inflate stream.
    Returns (compress s, decompress s, compressed bytes)."""
    import ctypes as C
    ver = lib.zlibVersion()
    st = _ZStream()
    t0 = time.perf_counter()
    assert lib.deflateInit2_(C.byref(st), 1, 8, -15, 9, 0, ver, C.sizeof(_ZStream)) == 0
    base_in, base_out = C.addressof(src_buf), C.addressof(dst_buf)
    pos = 0
    while True:
        send = min(CHUNK, n - pos)
        st.next_in = base_in + pos; st.avail_in = send
        st.next_out = base_out + st.total_out; st.avail_out = dst_cap - st.total_out
        pos += send
        fin = pos >= n
        rc = lib.deflate(C.byref(st), 4 if fin else 3)
        assert rc == (1 if fin else 0) and st.avail_in == 0, rc
        if fin:
            break
    clen = st.total_out
    lib.deflateEnd(C.byref(st))
    t1 = time.perf_counter()
    zi = _ZStream()
    assert lib.inflateInit2_(C.byref(zi), -15, ver, C.sizeof(_ZStream)) == 0
    zi.next_in = base_out; zi.avail_in = clen
    zi.next_out = C.addressof(back_buf); zi.avail_out = n
    rc = lib.inflate(C.byref(zi), 2)                                 # Z_SYNC_FLUSH, as the reference calls it
    assert rc == 1 and zi.total_out == n, (rc, zi.total_out)
    lib.inflateEnd(C.byref(zi))
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, clen


def cpu_worker(idx: int, shard_mb: int, data_path: str, cpu: int):
    """one worker process of the CPU baseline, pinned to one physical core the way the reference's script pins its
    processes with taskset (test/performance_tests/run_perf_test.sh:104-110): its own contiguous shard of the bench data
    (a file in shared memory the parent wrote), three passes of this host's libz through its C API and three of the
    software path's port (oracle/), all workers starting together; prints one JSON line"""
    import ctypes as C
    import numpy as np
    import oracle_lib as O
    if cpu >= 0 and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, {cpu})
        except OSError:
            pass
    base = np.memmap(data_path, dtype=np.uint8, mode="r")
    S = min(shard_mb << 20, len(base)) & ~(CHUNK - 1)
    span = max(1, len(base) - S)
    off = (idx * 7919 * CHUNK) % span
    src = bytes(base[off:off + S])
    del base
    lib = _libz()
    src_buf = C.create_string_buffer(src, S); cap = S * 9 // 8 + 65536
    dst_buf = C.create_string_buffer(cap); back_buf = C.create_string_buffer(S)
    O.sw_compress("GZIP_EXT", src[:CHUNK], CHUNK, 1)                 # tables built, pages touched
    if lib:
        _libz_pass(lib, src_buf, min(S, 4 * CHUNK), dst_buf, cap, back_buf)
    print("READY", flush=True)
    sys.stdin.readline()                                             # the parent lets every worker go at once
    res = {"n": S, "cpu": cpu, "port": [], "libz": []}
    for _ in range(3):
        if lib:
            tc, td, clen = _libz_pass(lib, src_buf, S, dst_buf, cap, back_buf)
            res["libz"].append([tc, td]); res["zclen"] = clen
    if lib:
        assert back_buf.raw == src
        res["zver"] = lib.zlibVersion().decode()
    for _ in range(3):
        t0 = time.perf_counter()
        rc, used, out, _ = O.sw_compress("GZIP_EXT", src, CHUNK, 1, cap=cap)
        t1 = time.perf_counter()
        rc2, _, back = O.sw_decompress("GZIP_EXT", out, S + 64)
        t2 = time.perf_counter()
        assert rc == 0 and used == S and rc2 == 0 and back == src
        res["port"].append([t1 - t0, t2 - t1]); res["clen"] = len(out)
    print(json.dumps(res), flush=True)


def cpu_baseline(shard_mb: int, base_mb: int, max_workers: int):
    """The reference's arithmetic on this box's host cores, the way the reference measures itself
    (test/performance_tests/run_perf_test.sh:100-123): N independent worker PROCESSES, one per physical core and pinned to
    it, each on its own contiguous shard (>= 64 MiB) of the bench buffer, all released together, three passes each; the
    per-worker rates of a pass summed, the MEDIAN pass reported with the spread of the three.  `value` is this host's own
    libz driven like src/qatzip_sw.c:178-231 / :323-351 when it is the pinned 1.2.11 (its bytes are the reference's:
    tests/test_oracle.py) - kind "libz"; the port of oracle/ (the checker, a slower restatement) stands beside it."""
    import subprocess
    import tempfile
    import datagen
    cpus, model = host_cpu()
    nw = max(1, min(len(cpus), max_workers))
    # a container's CFS quota is what the box really gives: 128 physical cores were visible on the MI355X box, cpu.max said
    # "1600000 100000" = 16 CPUs, and the summed rate stood still from 16 workers on (gpurun_out/r5l_cpu.log: 8 / 16 / 32 / 64 / 128
    # pinned workers 1.9 / 3.5 / 3.6 / 3.0 / 3.3 GB/s) - more workers than that only measure the scheduler's throttling
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    if quota is not None:
        nw = max(1, min(nw, int(quota)))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(shm, "qatzip_amd_bench_%d.bin" % os.getpid())
    datagen.gen("silesia", max(base_mb, shard_mb + 16) << 20, 20250523).tofile(path)

    def run(count):
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--cpu-mb", str(shard_mb),
                                "--cpu-data", path, "--cpu-pin", str(cpus[i % len(cpus)])],
                               stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(count)]
        ready = [p for p in ps if (p.stdout.readline() or "").startswith("READY")]
        for p in ready:
            try:
                p.stdin.write("GO\n"); p.stdin.flush()
            except OSError:
                pass
        out = []
        for p in ps:
            try:
                o, _ = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                p.kill(); continue
            line = [x for x in o.splitlines() if x.startswith("{")]
            if p.returncode == 0 and line:
                out.append(json.loads(line[-1]))
        return out
    try:
        one = run(1)
        many = run(nw)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    if not one or not many:
        return {"error": "CPU baseline workers failed"}

    def rates(rs, key):
        """per pass: (both directions, compress, decompress) GB/s summed over the workers; then the median pass and the spread"""
        rs = [r for r in rs if len(r.get(key, [])) == 3]
        if not rs:
            return None
        passes = []
        for k in range(3):
            passes.append((sum(2 * r["n"] / (r[key][k][0] + r[key][k][1]) for r in rs) / 1e9,
                           sum(r["n"] / r[key][k][0] for r in rs) / 1e9, sum(r["n"] / r[key][k][1] for r in rs) / 1e9))
        order = sorted(passes)
        med = order[1]
        return {"value": round(med[0], 4), "compress": round(med[1], 4), "decompress": round(med[2], 4),
                "spread": round((order[2][0] - order[0][0]) / med[0], 4), "passes": [round(p[0], 4) for p in passes], "workers": len(rs)}
    port, port1 = rates(many, "port"), rates(one, "port")
    libz, libz1 = rates(many, "libz"), rates(one, "libz")
    zver = next((r["zver"] for r in many if "zver" in r), None)
    pinned = libz is not None and zver == "1.2.11"
    head = libz if pinned else port
    cpu_s = sum(sum(a + b for a, b in r.get("port", [])) + sum(a + b for a, b in r.get("libz", [])) for r in many)
    res = {"value": head["value"], "unit": "GB/s", "cores": head["workers"], "kind": "libz" if pinned else "port",
           "compress": head["compress"], "decompress": head["decompress"], "spread": head["spread"], "passes": head["passes"],
           "one_worker": (libz1 if pinned else port1), "cpu_model": model, "physical_cores": len(cpus),
           "cgroup_cpu_quota": quota,
           "sample": ("the container's CFS quota is %.0f CPUs of the %d physical cores: " % (quota, len(cpus)) if quota is not None and quota < len(cpus) else "") +
                     "%d worker process(es), each pinned to one physical core, x %d MiB contiguous shards of the same buffer, "
                     "GZIP_EXT L1 64 KB chunks, compress + decompress, three passes started together, median pass of the "
                     "summed per-worker rates (%.0f s of CPU work in all)" % (len(many), many[0]["n"] >> 20, cpu_s),
           "ratio": round(sum(r.get("clen", 0) for r in many) / max(1, sum(r["n"] for r in many if "clen" in r)), 4),
           "port": dict(port or {}, one_worker=port1, note="oracle/libqzoracle.so: the CPU restatement the parity tests check against"),
           }
    if libz is not None:
        res["host_libz"] = dict(libz, zlibVersion=zver, one_worker=libz1,
                                note="deflateInit2(1, 8, -15, 9, 0) + deflate(Z_FULL_FLUSH) per 64 KB, inflate(Z_SYNC_FLUSH): the loops of "
                                     "src/qatzip_sw.c:178-231 and :323-351 through libz's C API (ctypes)")
    return res


# ------------------------------------------------------------------ extra legs (outside the timed region)
def api_leg(base, tile, mb, timed=3):
    """qzCompress / qzDecompress themselves, host to host on qzMalloc(PINNED_MEM) buffers: PCIe both ways included.  One pass
    that sizes the session's buffers, then `timed` passes: the median of each direction"""
    import ctypes as C
    from qatzip_amd import api as A
    n = mb << 20
    L = A.lib()
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, CHUNK)
    cap = L.qzMaxCompressedLength(n, C.byref(s.s)) + 64
    p_src, p_dst, p_back = L.qzMalloc(n, 0, A.PINNED_MEM), L.qzMalloc(cap, 0, A.PINNED_MEM), L.qzMalloc(n + 64, 0, A.PINNED_MEM)
    assert p_src and p_dst and p_back, "qzMalloc(PINNED_MEM) failed"
    hsrc = np.ctypeslib.as_array((C.c_ubyte * n).from_address(p_src))
    for off in range(0, n, tile):
        k = min(tile, n - off)
        hsrc[off:off + k] = base[:k]
    res = {}
    comp_len = 0
    tcs, tds = [], []
    for it in range(1 + timed):                                    # first pass sizes the session's buffers
        sl, dl = C.c_uint(n), C.c_uint(cap)
        t0 = time.perf_counter()
        rc = L.qzCompress(C.byref(s.s), C.cast(p_src, C.c_char_p), C.byref(sl), p_dst, C.byref(dl), 1)
        tc = time.perf_counter() - t0
        assert rc == 0 and sl.value == n, rc
        comp_len = dl.value
        sl2, dl2 = C.c_uint(comp_len), C.c_uint(n + 64)
        t0 = time.perf_counter()
        rc = L.qzDecompress(C.byref(s.s), C.cast(p_dst, C.c_char_p), C.byref(sl2), p_back, C.byref(dl2))
        td = time.perf_counter() - t0
        assert rc == 0 and dl2.value == n, rc
        if it:
            tcs.append(tc); tds.append(td)
    tc, td = sorted(tcs)[len(tcs) // 2], sorted(tds)[len(tds) // 2]
    back = np.ctypeslib.as_array((C.c_ubyte * n).from_address(p_back))
    assert np.array_equal(back[:1 << 20], hsrc[:1 << 20]) and np.array_equal(back[-(1 << 20):], hsrc[-(1 << 20):])
    res = {"api_bytes_MiB": mb, "api_compress_GBps": round(n / tc / 1e9, 3), "api_decompress_GBps": round(n / td / 1e9, 3),
           "api_decompress_ms": [round(t * 1e3, 2) for t in tds],
           "api_pinned": int(L.qzMemFindAddr(p_src + 12345)), "api_note": "one qzCompress + one qzDecompress call, host to host, "
           "qzMalloc(PINNED_MEM) source and destination, PCIe included; the median of %d passes each" % timed}
    for p in (p_src, p_dst, p_back):
        L.qzFree(p)
    s.close()
    return res


def src_sha256():
    """SHA-256 over the kernel and host sources of the library (qatzip_amd/csrc, names and contents): what a committed profile
    was taken from - a PMC figure is only quoted for the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "qatzip_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_traffic(fname, command_has, kernels, key="hbm_bytes_fetch_x2"):
    """HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction) of the kernels whose names start
    with one of `kernels`, summed, from the committed rocprofv3 --pmc summary profiles/<fname> (tools/pmc_summary.py) - or
    None when the file is missing, was taken from other sources than the ones that are running (src_sha256) or from another
    command.  Never stale: null rather than a number measured on different code."""
    try:
        with open(os.path.join(ROOT, "profiles", fname)) as f:
            pj = json.load(f)
        if pj.get("src_sha256") != src_sha256() or not all(w in pj.get("command", "") for w in command_has):
            return None
        tot = 0
        for pre in kernels:
            ks = [k for k in pj["kernels"] if k.startswith(pre)]
            if not ks:
                return None
            tot += pj["kernels"][max(ks, key=lambda k: pj["kernels"][k]["hbm_bytes_fetch_x2"])][key]   # the whole-call launch
        return tot
    except (OSError, KeyError, ValueError):
        return None


def raw_sweep(ctx, qz, d_src, mb, sizes=(16384, 65536, 131072)):
    """BASELINE config 3: raw deflate + raw inflate, hw_buff_sz 16 / 64 / 128 KB, data resident in HBM.  Beside the call
    rates: the kernels' own time by HIP events (K1 with K2 and the CRC in its waves; the inflate kernels), the fraction of
    the HBM peak their algorithmic bytes (U + C) make of it, and the HBM traffic of the committed PMC passes of this leg
    (profiles/r4_raw<K>_pmc.json, null when they were taken from other code)."""
    n = mb << 20
    out = {}
    d_c = ctx.alloc(qz.max_deflate_len(n, 16384))
    d_b = ctx.alloc(n + 4096)
    for hw in sizes:
        best_c = best_d = None
        for _ in range(2):
            ctx.k1_stats(reset=True)
            ctx.sync(); t0 = time.perf_counter()
            ctx.deflate_raw_async(view(qz, d_src, 0, n), n, hw, 1, 1, d_c); ctx.sync()
            t1 = time.perf_counter()
            cl = ctx.result()
            k1_ms, k1_l, _ = ctx.k1_stats()
            t2 = time.perf_counter()
            iu, ol, _ = ctx.inflate_stream(d_c, cl, d_b, hw, want_crc=False)
            t3 = time.perf_counter()
            assert iu == cl and ol == n
            inf = ctx.inflate_timing()
            if best_c is None or t1 - t0 < best_c:
                best_c, kc = t1 - t0, k1_ms
            if best_d is None or t3 - t2 < best_d:
                best_d, kd = t3 - t2, (inf[3] + inf[2]) if inf[3] > 0 else inf[0]
        k1 = ctx.timing()
        tag = "%dK" % (hw >> 10)
        alg = float(n + cl)
        out[tag] = {"compress_GBps": round(n / best_c / 1e9, 2), "decompress_GBps": round(n / best_d / 1e9, 2),
                    "ratio": round(cl / n, 4), "lz77_ms_first_batch": round(k1[0], 2),
                    "deflate_kernel_ms": round(kc, 2), "deflate_frac": round(alg / (kc * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kc > 0 else None,
                    "inflate_kernel_ms": round(kd, 2), "inflate_frac": round(alg / (kd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kd > 0 else None,
                    "deflate_traffic": pmc_traffic("r6_raw%d_pmc.json" % (hw >> 10), ("legs_run.py raw%d %d" % (hw >> 10, mb),), ("qzk_lz77_pull_kernel",)),
                    "inflate_traffic": pmc_traffic("r6_raw%d_pmc.json" % (hw >> 10), ("legs_run.py raw%d %d" % (hw >> 10, mb),),
                                                   ("qzk_inflate_spec_kernel", "qzk_lz_resolve_kernel"))}
    assert ctx.crc32(d_b, n) == ctx.crc32(view(qz, d_src, 0, n), n)
    d_c.free(); d_b.free()
    out["bytes_MiB"] = mb
    return out


def lz4_leg(ctx, qz, d_src, mb):
    """BASELINE config 4: one LZ4 frame (one 64 KB block, content size + XXH32) per 64 KB, compress then decompress"""
    n = mb << 20
    nfr = n // CHUNK
    d_c = ctx.alloc(n + nfr * 64 + 4096)
    d_b = ctx.alloc(n + 4096)
    best_c = best_d = None
    for _ in range(2):
        t0 = time.perf_counter()
        cl, lens = ctx.lz4_compress_frames(view(qz, d_src, 0, n), n, d_c, CHUNK)
        t1 = time.perf_counter()
        kc = ctx.timing()[3]                                            # the call's kernels, first to last, by HIP events
        offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]])
        segs = np.zeros(nfr, qz._lib.LZ4SEG_DT)
        segs["in_off"] = offs; segs["out_off"] = np.arange(nfr, dtype=np.int64) * CHUNK; segs["in_len"] = lens; segs["out_cap"] = CHUNK
        res = np.zeros(nfr, qz._lib.LZ4RES_DT)
        t2 = time.perf_counter()
        ctx._chk(ctx.L.qzd_lz4_decompress_frames(ctx.h, d_c.ptr, d_b.ptr, segs.ctypes.data, nfr, res.ctypes.data))
        t3 = time.perf_counter()
        kd = ctx.timing()[3]
        assert (res["status"] == 0).all() and (res["out_len"] == CHUNK).all()
        best_c = min(best_c or 1e9, t1 - t0); best_d = min(best_d or 1e9, t3 - t2)
    assert ctx.crc32(d_b, n) == ctx.crc32(view(qz, d_src, 0, n), n)
    d_c.free(); d_b.free()
    alg = float(n + cl)
    return {"bytes_MiB": mb, "compress_GBps": round(n / best_c / 1e9, 2), "decompress_GBps": round(n / best_d / 1e9, 2),
            "ratio": round(cl / n, 4),
            "compress_kernel_ms": round(kc, 2), "compress_frac": round(alg / (kc * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kc > 0 else None,
            "decompress_kernel_ms": round(kd, 2), "decompress_frac": round(alg / (kd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kd > 0 else None,
            "compress_traffic": pmc_traffic("r6_lz4_pmc.json", ("legs_run.py lz4 %d" % mb,), ("qzk_lz4c_pull_kernel",)),
            "decompress_traffic": pmc_traffic("r6_lz4_pmc.json", ("legs_run.py lz4 %d" % mb,), ("qzk_lz4d_kernel",)),
            "note": "64 KB frames, XXH32 content checksum made and verified in-kernel; call rates host call to host return, "
                    "kernel_ms / frac = (U + C) over the kernels' HIP-event time over the HBM peak"}


def sessions_leg(qz, dev, d_src, total, steps):
    """The same job driven the way the reference's harness drives a device (test/main.c:2175-2202, `-t`): one session per
    host thread, started together.  Thread i owns call i of the buffer: compress it, decompress it, `steps` times.  Not the
    headline (that is one session, one call after the other); it shows what the device does when calls overlap - a call's
    phase A is one latency chain per segment and leaves the chip half empty, a second session's kernels fill it."""
    import threading
    call_n = [min(API_CALL_BYTES, total - o) for o in range(0, total, API_CALL_BYTES)]
    ctxs = [qz.Context(dev) for _ in call_n]
    backs = [c.alloc(n) for c, n in zip(ctxs, call_n)]
    d_comp = [c.alloc(qz.max_deflate_len(n, CHUNK)) for c, n in zip(ctxs, call_n)]
    err = []
    bar = threading.Barrier(len(call_n) + 1)

    def body(i):
        c, n = ctxs[i], call_n[i]
        try:
            src = view(qz, d_src, i * API_CALL_BYTES, n)
            for it in range(steps + 1):                 # pass 0 warms this context up (scratch, tables)
                if it == 1:
                    bar.wait(); bar.wait()
                c.deflate_raw_async(src, n, CHUNK, 1, 1, d_comp[i]); c.sync()
                cl = c.result()
                iu, ol, _ = c.inflate_stream(d_comp[i], cl, backs[i], CHUNK, want_crc=True)
                assert iu == cl and ol == n
        except Exception as e:   # noqa: BLE001
            err.append(repr(e))
            try:
                bar.abort()
            except Exception:   # noqa: BLE001
                pass

    th = [threading.Thread(target=body, args=(i,)) for i in range(len(call_n))]
    for t in th:
        t.start()
    try:
        bar.wait()                                      # everybody warmed up
        t0 = time.perf_counter()
        bar.wait()
    except threading.BrokenBarrierError:
        pass
    for t in th:
        t.join()
    dt = time.perf_counter() - t0 if not err else 0.0
    for b in backs + d_comp:
        b.free()
    for c in ctxs:
        c.close()                                       # the contexts' decode scratch (several GiB each) goes back
    if err or dt <= 0:
        return {"error": "; ".join(err)[:200]}
    return {"sessions": len(call_n), "GBps": round(2.0 * sum(call_n) * steps / dt / 1e9, 3), "steps": steps,
            "note": "the same buffer as calls of 2 GiB (what a qatzip.h caller makes of it), one host thread and one session per call, started together; compress + decompress, uncompressed bytes both ways"}


def view(qz, buf, off, n):
    v = qz.DevBuf.__new__(qz.DevBuf)
    v.ctx, v.nbytes, v.ptr = buf.ctx, n, buf.ptr + off
    return v


def one_stream_leg(ctx, qz, pg, rank, world, d_src, slice_mb, members, steps, ndev=None, progress=None):
    """BASELINE config 5 as written, timed: a logical buffer of `members` x (world x slice_mb MiB) - 16 members of 8 x 511 MiB
    = 64 GB on an 8-GPU node; a gzip-ext header describes less than 4 GiB, so the buffer is a SEQUENCE of members - dealt to
    the ranks like a striped volume (shard.member_plan).  Every member is built by all ranks: each deflates its shard on its
    GPU, the compressed shards travel to rank 0's HBM, rank 0 folds the CRCs, closes the member and appends it to the output;
    member m + 1 is deflated while member m's shards are on the wire (shard.OneStream.run_members: the transport has its
    own context and thread, two staging buffers).  `steps` passes per transport between barriers, max-over-ranks time,
    deflate included.  Both transports are measured (IPC window with peer copies / RCCL all-gather + send-recv group) and the
    faster one is the leg's headline; a transport that cannot start on this box reports its error (and RCCL's own log)
    instead.  Rank 0 checks, outside the timed loop: every member's header and trailer against the plan and against the
    ranks' own CPU CRC-32s of member 0, and a prefix of member 0's payload against the oracle."""
    import zlib
    from qatzip_amd import shard
    t_leg = time.perf_counter()

    def note(what):                                                  # where the leg stands: for the line of a run that timed out ...
        if progress:
            progress(what)
        if os.environ.get("QATZIP_AMD_BENCH_TRACE"):                 # ... and, rank by rank, for whoever debugs one
            print("[one-stream leg, rank %d, +%.1f s] %s" % (rank, time.perf_counter() - t_leg, what), file=sys.stderr, flush=True)
    # the ranks of this leg share one node by contract: RCCL's bootstrap sockets may use the loopback interface (the
    # container's hostname need not resolve); a launcher that knows better sets the variable itself
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    rccl_log = "/tmp/qatzip_amd_rccl_%d.log" % os.getpid()
    os.environ.setdefault("NCCL_DEBUG", "WARN"); os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
    sl = slice_mb << 20
    total = members * world * sl
    plan = shard.member_plan(total, world, CHUNK, sl)
    mine = shard.local_offsets(plan, rank)
    need = sum(n for _, n in mine)
    # this rank's shards, back to back: the bench buffer, repeated when the plan is longer than it (synthetic either way).
    # What can fail on ONE rank here (device memory: eight ranks rehearsing on one GPU share its HBM) is agreed upon before any
    # transport is set up - a rank that left alone would leave the others waiting in the transport's first broadcast
    note("buffers: %d MiB of shards per rank" % (need >> 20))
    own = src = host = d_out = None
    n0 = mine[0][1]
    prep_err = None
    try:
        if need <= d_src.nbytes:
            src = view(qz, d_src, 0, need)
        else:
            own = src = ctx.alloc(need)
            for off in range(0, need, d_src.nbytes):
                ctx._chk(ctx.L.qzd_d2d(ctx.h, src.ptr + off, d_src.ptr, min(d_src.nbytes, need - off)))
        host = src.download(n0) if n0 <= (1 << 30) else None
        d_out = ctx.alloc(total // 2 + (1 << 26)) if rank == 0 else None      # the bench data compresses to < 0.4: half is room enough
    except Exception as e:   # noqa: BLE001
        prep_err = "rank %d could not prepare its buffers (%d MiB of shards%s): %s" % (
            rank, need >> 20, ", %d MiB of output" % ((total // 2 + (1 << 26)) >> 20) if rank == 0 else "", str(e)[:120])
    my_crc = zlib.crc32(host.tobytes()) & 0xffffffff if host is not None else 0
    out = {"ranks": world, "members": len(plan), "slice_MiB": slice_mb, "member_raw_MiB": world * slice_mb,
           "buffer_GB": round(total / 1e9, 2), "steps": steps}
    if not shard._all_ok(pg, prep_err is None):
        for b in (own, d_out):
            if b is not None:
                b.free()
        out["error"] = prep_err or "another rank could not prepare its buffers (device memory shared by %d ranks?)" % world
        return out
    best = None
    for transport in ("ipc", "rccl"):
        if transport == "rccl" and ndev is not None and ndev < world:
            # a rehearsal of N ranks on fewer devices (a one-GPU box): RCCL admits ONE rank per device (ncclCommInitRank:
            # "invalid usage" otherwise) - its transport is not started, and the line says so instead of printing RCCL's error
            out["rccl"] = {"skipped": "%d ranks on %d device(s): RCCL takes one rank per GPU; the IPC window carries this run, the RCCL "
                                      "transport (ncclAllGather + ncclSend/ncclRecv group, qzd_rccl_*) needs one GPU per rank" % (world, ndev)}
            continue
        try:
            note("%s: setting the transport up" % transport)
            one = shard.OneStream(ctx, pg, rank, world, sl, CHUNK, 1, transport)
            if one.error:
                out[transport] = {"error": one.error[:200]}
                continue
            note("%s: first pass over %d members (warm-up, checked)" % (transport, len(plan)))
            r = one.run_members(src, plan, d_out)                    # warm-up + the pass that is checked
            if "error" in r:
                out[transport] = {"error": r["error"][:200]}
                one.close()
                continue
            recs = shard.all_gather_records(pg, shard.pack_record(n0, 0, my_crc), world) if world > 1 else [(n0, 0, my_crc)]
            verified = None
            if rank == 0:
                verified = r["raw_bytes"] == total and len(r["member_bytes"]) == len(plan)
                pos = 0
                for m, mb_len in enumerate(r["member_bytes"]):         # every member: header sizes and ISIZE against the plan
                    raw_m = sum(n for _, n in plan[m])
                    hdr = d_out.download(24, pos).tobytes(); tr = d_out.download(8, pos + mb_len - 8).tobytes()
                    verified = verified and hdr[:4] == b"\x1f\x8b\x08\x04" and hdr[12:14] == b"QZ" and \
                        int.from_bytes(hdr[16:20], "little") == raw_m and int.from_bytes(hdr[20:24], "little") == mb_len - 32 and \
                        int.from_bytes(tr[4:], "little") == raw_m & 0xffffffff
                    if m == 0 and host is not None:                    # member 0: its CRC-32 from the ranks' own, a prefix against the oracle
                        crc = 0
                        for i, (rl, _, c) in enumerate(recs):
                            crc = c if i == 0 else shard.crc32_combine(crc, c, rl)
                        verified = verified and int.from_bytes(tr[:4], "little") == crc
                        import oracle_lib as O
                        k = min(n0, 4 << 20) // CHUNK * CHUNK
                        exp = O.sw_compress("RAW", host[:k].tobytes(), CHUNK, 1, last=0 if (world > 1 or k < n0) else 1, cap=k * 9 // 8 + 65536)[2]
                        verified = verified and d_out.download(len(exp), 24).tobytes() == exp
                    pos += mb_len
            note("%s: checked, timed passes" % transport)
            barrier(pg)
            t0 = time.perf_counter()
            tg = td = tov = 0.0
            ok = True
            for _ in range(steps):
                r2 = one.run_members(src, plan, d_out)
                ok = ok and "error" not in r2
                tg += r2.get("gather_ms", 0.0); td += r2.get("deflate_ms", 0.0); tov += r2.get("overlapped_ms", 0.0)
            barrier(pg)
            dt = allreduce(pg, time.perf_counter() - t0, "MAX")
            tg = allreduce(pg, tg, "MAX"); tov = allreduce(pg, tov, "MAX"); td = allreduce(pg, td, "MAX")
            one.close()
            if not ok:
                out[transport] = {"error": "a member of the timed loop failed"}
                continue
            res = {"GBps": round(total * steps / dt / 1e9, 3), "ms_per_pass": round(dt / steps * 1e3, 2),
                   "ms_per_member": round(dt / steps / len(plan) * 1e3, 2), "deflate_ms_per_member": round(td / steps / len(plan), 2),
                   "gather_ms_per_member": round(tg / steps / len(plan), 2),
                   "gather_ms_beside_a_deflate_per_member": round(tov / steps / len(plan), 2),
                   "gather_share": round(max(0.0, tg - tov) / steps / (dt / steps * 1e3), 3)}
            if rank == 0:
                res.update({"out_bytes": r["out_bytes"], "verified": bool(verified)})
            out[transport] = res
            if best is None or res["GBps"] > out[best]["GBps"]:
                best = transport
        except Exception as e:   # noqa: BLE001 - a transport that does not work here must not cost the other one
            out[transport] = {"error": repr(e)[:200]}
        if transport == "rccl" and "error" in out.get("rccl", {}):
            try:                                                     # what RCCL itself had to say (NCCL_DEBUG=WARN into a file of ours)
                with open(os.environ.get("NCCL_DEBUG_FILE", rccl_log)) as f:
                    lines = [ln.strip() for ln in f if ln.strip()]
                out["rccl"]["rccl_log"] = " | ".join(lines[-3:])[:400]
            except OSError:
                pass
    out["transport"] = best
    if best:
        out["GBps"] = out[best]["GBps"]
        out["overlapped"] = out[best]["gather_ms_beside_a_deflate_per_member"] > 0
    out["note"] = ("ipc = peer copies into an IPC window in rank 0's HBM (xGMI between GPUs); rccl = ncclAllGather of the records + "
                   "ncclSend/ncclRecv group; uncompressed bytes of all members / max-over-ranks time, deflate included; "
                   "gather_share = the part of a pass its gathers take that no deflate ran beside")
    if own is not None:
        own.free()
    if d_out is not None:
        d_out.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mb", type=int, default=4096, help="buffer size per GPU in MiB (default: the 4 GB config)")
    ap.add_argument("--base-mb", type=int, default=128, help="distinct synthetic data generated per GPU (tiled)")
    ap.add_argument("--cpu-mb", type=int, default=64, help="CPU baseline: shard size per worker, MiB")
    ap.add_argument("--cpu-data", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-pin", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=256, help="CPU baseline: at most this many worker processes")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the API / RAW sweep / LZ4 / one-stream legs")
    ap.add_argument("--extra-mb", type=int, default=1024, help="bytes of the extra legs, MiB")
    ap.add_argument("--api-mb", type=int, default=2047, help="bytes of the one qzCompress / qzDecompress call of the API leg, MiB "
                    "(qatzip.h lengths are 32-bit)")
    ap.add_argument("--members", type=int, default=0, help="members of the one-stream leg (multi-rank runs; default 16 from 8 ranks on, else 4)")
    ap.add_argument("--no-probe", action="store_true", help="skip the lone 12288-chunk K1 launch after the timed region "
                    "(the rocprofv3 --pmc passes: every K1 launch of the run is then a whole 2 GiB call)")
    args = ap.parse_args()
    if args.cpu_worker is not None:
        cpu_worker(args.cpu_worker, args.cpu_mb, args.cpu_data, args.cpu_pin)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                      # does not return

    rank, world, local, pg = dist_setup()
    import datagen
    import qatzip_amd

    # one process per GPU: LOCAL_RANK picks the device.  More ranks than GPUs (exercising the multi-rank path on a smaller
    # box) wrap around and the line says so; QATZIP_AMD_BENCH_DEVICE pins every rank to one device.
    ndev = max(1, qatzip_amd.load().qzd_device_count())
    dev = int(os.environ.get("QATZIP_AMD_BENCH_DEVICE", local % ndev))
    ctx = qatzip_amd.Context(dev)
    total = args.mb << 20
    base_n = min(args.base_mb << 20, total)
    base = datagen.gen("silesia", base_n, 20250523 + rank)
    d_src = ctx.alloc(total)
    # tile with a period that is not a multiple of the chunk size: every copy of the distinct data is cut into chunks at
    # different places, so no two chunks of the buffer are equal (and the decoder cannot profit from duplicates)
    tile = base_n - TILE_SKEW if total > base_n else base_n
    for off in range(0, total, tile):
        d_src.upload(base[:min(tile, total - off)], off)
    ncalls = (total + CALL_BYTES - 1) // CALL_BYTES
    call_n = [min(CALL_BYTES, total - i * CALL_BYTES) for i in range(ncalls)]
    d_comp = [ctx.alloc(qatzip_amd.max_deflate_len(n, CHUNK)) for n in call_n]
    d_back = ctx.alloc(max(call_n))

    comp_len = [0] * ncalls

    def compress_all():
        for i, n in enumerate(call_n):
            ctx.deflate_raw_async(view(qatzip_amd, d_src, i * CALL_BYTES, n), n, CHUNK, 1, 1, d_comp[i])
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
    assert ctx.crc32(d_back, call_n[-1]) == ctx.crc32(view(qatzip_amd, d_src, (ncalls - 1) * CALL_BYTES, call_n[-1]), call_n[-1])

    ctx.k1_stats(reset=True)
    barrier(pg); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync(); barrier(pg)
    dt = allreduce(pg, time.perf_counter() - t0, "MAX")
    # dominant kernel (K1: the LZ77 parse, with K2 and the CRC in its waves): HIP events around every launch of the timed
    # region, on the stream it is launched on
    k1_ms, k1_launches, k1_chunks = ctx.k1_stats()

    # per-direction split (untimed by the contract, reported alongside) and one K1 launch alone on the chip
    ctx.sync(); t1 = time.perf_counter(); compress_all(); ctx.sync(); tc = time.perf_counter() - t1
    t1 = time.perf_counter(); decompress_all(); ctx.sync(); td = time.perf_counter() - t1
    inf_ms = ctx.inflate_timing()
    batch_chunks = ctx.batch_chunks()         # three rounds over the resident waves: the probe launch below
    probe_n = min(batch_chunks * CHUNK, call_n[0])
    if not args.no_probe:
        ctx.deflate_raw_async(view(qatzip_amd, d_src, 0, probe_n), probe_n, CHUNK, 1, 1, d_comp[0]); ctx.sync()
    k_ms = ctx.timing()                      # the probe: one launch of three rounds over the resident waves
    comp_total = allreduce(pg, float(sum(comp_len)), "SUM")
    raw_total = float(total) * world
    from qatzip_amd import shard as _shard
    per_rank = _shard.allgather_floats(pg, [tc, td])                 # every rank's own compress / decompress pass (seconds)
    tc = allreduce(pg, tc, "MAX"); td = allreduce(pg, td, "MAX")

    extra = {}
    one = None
    hard_exit = False
    if not args.no_extra:
        if world > 1:
            # The leg runs on a helper thread with a deadline: its transports wait for other ranks on the device (RCCL) or
            # in bounded polls (IPC window), and a rank that fell out of step must not take the headline with it - the
            # timed region is already reduced over the ranks at this point.  After a timeout no further collective runs.
            import threading
            box = {}

            def leg():
                try:
                    # 8 ranks x 511 MiB: the largest member a gzip-ext header can describe (both sizes are 32-bit); sixteen of
                    # them are BASELINE config 5's 64 GB.  Fewer ranks (a one-GPU box exercising the path): four members.
                    for d in d_comp:
                        d.free()
                    d_back.free()
                    members = args.members or (16 if world >= 8 else 4)
                    box["one"] = one_stream_leg(ctx, qatzip_amd, pg, rank, world, d_src, min(511, 4095 // world, args.mb), members, max(1, args.steps), ndev,
                                                lambda what: box.__setitem__("at", what))
                except Exception as e:   # noqa: BLE001 - the headline must survive a box without peer access
                    box["one"] = {"error": str(e)[:200]}
            th = threading.Thread(target=leg, daemon=True)
            th.start()
            th.join(float(os.environ.get("QATZIP_AMD_BENCH_LEG_TIMEOUT", "240")))
            if th.is_alive():
                one = {"error": "the one-stream leg did not finish in %s s on rank %d; it was at: %s" %
                                (os.environ.get("QATZIP_AMD_BENCH_LEG_TIMEOUT", "240"), rank, box.get("at", "its preparation (buffers, member 0's CRC)"))}
                hard_exit = True
            else:
                one = box.get("one")
            hard_exit = True        # multi-rank runs end without a last rendezvous: rank 0's line never waits for a peer's teardown
        elif rank == 0:
            emb = min(args.extra_mb, args.mb)
            for d in d_comp:
                d.free()
            d_back.free()
            if total > API_CALL_BYTES:
                try:
                    extra["concurrent_sessions"] = sessions_leg(qatzip_amd, dev, d_src, total, args.steps)
                except Exception as e:   # noqa: BLE001 - an extra leg must not cost the headline
                    extra["concurrent_sessions"] = {"error": str(e)[:200]}
            extra["hbm_copy_GBps"] = round(ctx.stream_copy_peak(1 << 30, 3), 1)
            try:
                h2d, d2h = ctx.pcie_peak(1 << 30, 2)
                extra["pcie_h2d_GBps"], extra["pcie_d2h_GBps"] = round(h2d, 2), round(d2h, 2)
            except Exception as e:   # noqa: BLE001
                extra["pcie_error"] = str(e)[:120]
            extra["raw_sweep"] = raw_sweep(ctx, qatzip_amd, d_src, emb)
            extra["lz4"] = lz4_leg(ctx, qatzip_amd, d_src, emb)
            extra.update(api_leg(base, tile, min(args.api_mb, args.mb)))

    if rank == 0:
        # HBM traffic per K1 launch: PMC counters cannot be read from inside this process; they come from the committed
        # rocprofv3 --pmc passes of this same command (profiles/r4_pmc.json, tools/profile_round.sh + tools/pmc_summary.py:
        # FETCH_SIZE and WRITE_SIZE collected in separate runs, KiB -> bytes, FETCH x2 per the gfx950 note) and are quoted
        # only when that file was taken from the sources that are running now (SHA-256 over qatzip_amd/csrc) with this
        # command's --mb - null otherwise, never stale.
        traffic = pmc_traffic("r6_pmc.json", ("bench.py --mb %d " % args.mb,), ("qzk_lz77_pull_kernel",))
        traffic_dec = pmc_traffic("r6_pmc.json", ("bench.py --mb %d " % args.mb,), ("qzk_inflate_spec_kernel", "qzk_lz_resolve_kernel"))
        traffic_raw = pmc_traffic("r6_pmc.json", ("bench.py --mb %d " % args.mb,), ("qzk_lz77_pull_kernel",), key="hbm_bytes_raw")
        value = 2.0 * raw_total * args.steps / dt / 1e9
        ratio = comp_total / raw_total
        # algorithmic bytes of the K1 launches of the timed region: every input byte read once, every compressed byte
        # written once (U + C per chunk), divided over the launches; rank 0's own launches and time
        alg_total = args.steps * (float(total) + float(sum(comp_len)))
        alg_bytes = alg_total / max(k1_launches, 1)
        launch_ms = k1_ms / max(k1_launches, 1)
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        copy_peak = extra.get("hbm_copy_GBps")
        res = {
            "metric": "compress + decompress GB/s (input bytes), QZ_DEFLATE_GZIP_EXT L1, 64 KB chunks",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "QZ_DEFLATE_GZIP_EXT level 1, 64 KB chunks, %d MiB Silesia-like buffer per GPU "
                                   "(%d MiB distinct, tiled with period %d B so that no two chunks are equal), %d device-layer call(s) of "
                                   "<= 4 GiB per direction, compress then decompress" % (args.mb, base_n >> 20, tile, ncalls),
                       "chunk": CHUNK, "ratio": round(ratio, 4), "parallelism": "chunks sharded over %d rank(s), "
                       "no data-path collective in the timed region" % world,
                       "compress_GBps": round(raw_total / tc / 1e9, 3), "decompress_GBps": round(raw_total / td / 1e9, 3)},
            "roofline": {"bound": "hbm", "kernel": "qzk_lz77_pull_kernel", "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "algorithmic_bytes": int(alg_bytes),
                         # both readings of the counters: `traffic` = FETCH_SIZE x 2 + WRITE_SIZE (the guide's gfx950 correction, calibrated on
                         # wide streaming reads), `traffic_raw` = FETCH_SIZE + WRITE_SIZE as counted.  For this kernel's scattered 16-byte
                         # requests x 2 over-corrects (at the launch's duration it would be more than the copy rate measured on the box);
                         # the truth lies between.  What the bytes are: table gathers (a 16-byte entry per 64-byte line, L2 hit rate
                         # 19 %), table stores, the far candidates' sixteen bytes and the input (profiles/r6_k1_occupancy.txt)
                         "traffic_raw": traffic_raw,
                         "peak_measured_copy": copy_peak,
                         "frac_of_measured_copy": round(achieved / copy_peak, 6) if copy_peak else None,
                         "launch_ms": round(launch_ms, 3), "launches": int(k1_launches),
                         "chunks_per_launch": round(k1_chunks / max(k1_launches, 1), 1),
                         "full_launch_alone_ms": round(k_ms[0], 3), "full_launch_chunks": probe_n // CHUNK,
                         "fused": "K2 (Huffman coding) and the chunk CRC-32 run inside this kernel's waves",
                         # where this design stops (profiles/r5_k1_without_k2.txt): the same launch with K2
                         # compiled out - the parse and the CRC alone, measured once on an MI355X, not by this run
                         "design_ceiling_GBps": {"value": 57.8, "compress_input_GBps": 41.7, "launch_ms_4GiB": 103.0,
                                                 "what": "qzk_lz77_pull_kernel with K2 compiled out (-DQZK_K1_NOK2), 4 GiB, (U + C) / launch, on the round's launch shape "
                                                         "(4 waves x 5 workgroups a CU): K2 in the wave costs 3.2 of 106.2 ms (12.6 of 114.5 at round 5's sixteen waves); the parse "
                                                         "alone is no faster with twenty waves than with sixteen - it moves 3.5 TB/s of random 64-byte requests (table gathers "
                                                         "and stores, far candidates; a streaming copy reaches 4.6 on the box)",
                                                 "source": "profiles/r6_k1_occupancy.txt, profiles/r5_k1_without_k2.txt, profiles/r6_k1_experiments.txt"},
                         "other_kernels_ms": {"separate K2 / CRC launches": round(k_ms[1], 3), "scan+gather": round(k_ms[2], 3),
                                              "inflate kernels (last call)": round(inf_ms[0], 3),
                                              "of which qzk_lz_resolve_kernel": round(inf_ms[2], 3),
                                              "qzk_crc_kernel(last call)": round(inf_ms[1], 3)}},
        }
        # the decode side's kernels of the last call (phase A + phase B of the two-phase inflate, or the wave-per-segment
        # kernel), by the same rule: algorithmic bytes (compressed bytes read + plain bytes written) / their HIP-event time
        dec_alg = float(comp_len[-1]) + float(call_n[-1])
        dec_ms = inf_ms[3] + inf_ms[2] if inf_ms[3] > 0 else inf_ms[0]          # phase A + phase B kernels (or the wave-per-segment kernel)
        if dec_ms > 0:
            res["roofline_decode"] = {"bound": "hbm", "kernel": "qzk_inflate_spec_kernel<K> (phase A: Huffman decoding, K lanes per segment) + "
                                                                "qzk_lz_resolve_kernel (phase B: the batch engine of qzk_lz_batch.h)",
                                      "achieved": round(dec_alg / (dec_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(dec_alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": traffic_dec,
                                      "algorithmic_bytes": int(dec_alg), "launch_ms": round(dec_ms, 3),
                                      "phase_A_ms": round(inf_ms[3], 3), "phase_B_ms": round(inf_ms[2], 3),
                                      "inflate_step_ms_with_host_walk": round(inf_ms[0], 3), "call_MiB": call_n[-1] >> 20}
        res["config"].update(extra)
        if "pcie_h2d_GBps" in extra and "api_compress_GBps" in extra:
            # the host-to-host API against what bounds it: input over the link while the kernels run, output back
            kc, kd = res["config"]["compress_GBps"], res["config"]["decompress_GBps"]
            res["config"]["api_compress_vs_bound"] = round(extra["api_compress_GBps"] / min(extra["pcie_h2d_GBps"], kc), 3)
            res["config"]["api_decompress_vs_bound"] = round(extra["api_decompress_GBps"] / min(extra["pcie_d2h_GBps"], kd), 3)
        if ndev < world:
            res["config"]["ranks_share_devices"] = "%d ranks on %d device(s)" % (world, ndev)
        if one is not None:
            res["config"]["one_stream"] = one
        if world > 1:
            # what a SCALE record is read for, at the top level: every rank's own rates (its 4 GiB per direction) and what
            # the multi-GPU member costs (the gathers' share of a pass that no deflate ran beside)
            res["per_rank"] = {"compress_GBps": [round(total / r[0] / 1e9, 2) for r in per_rank],
                               "decompress_GBps": [round(total / r[1] / 1e9, 2) for r in per_rank]}
            if isinstance(one, dict):
                for tr in ("ipc", "rccl"):
                    leg = one.get(tr) if isinstance(one.get(tr), dict) else None
                    if leg and "gather_share" in leg:
                        res.setdefault("one_stream_gather_share", {})[tr] = leg["gather_share"]
                        res.setdefault("one_stream_GBps", {})[tr] = leg.get("GBps", leg.get("value"))
        if not args.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.cpu_mb, args.base_mb, args.cpu_threads)
        print(json.dumps(res), flush=True)
    if hard_exit:
        os._exit(0)                                  # a helper thread is still inside a collective: no orderly teardown
    if pg is not None:
        pg.destroy_process_group()


if __name__ == "__main__":
    main()
