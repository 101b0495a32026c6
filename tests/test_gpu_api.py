"""GPU parity through the qatzip.h C ABI: every deflate wire format bit-identical to the oracle's
restatement of the software path, stream continuation (last=0/1), error behaviour, the streaming
adapter and a plain-C caller (drop-in proof)."""
import ctypes as C
import os
import subprocess
import zlib

import pytest

import datagen
import oracle_lib as O
from qatzip_amd import api as A

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FMT = {"4B": A.QZ_DEFLATE_4B, "GZIP": A.QZ_DEFLATE_GZIP, "GZIP_EXT": A.QZ_DEFLATE_GZIP_EXT, "RAW": A.QZ_DEFLATE_RAW}


@pytest.mark.parametrize("fmt", ["GZIP_EXT", "GZIP", "RAW", "4B"])
def test_compress_matches_sw_path_bit_for_bit(fmt):
    for hw in (65536, 16384, 131072):
        s = A.Session(FMT[fmt], hw)
        assert s.rc_setup == A.QZ_OK
        for kind, n in (("silesia", 200777), ("text", 65536), ("rand", 70000), ("runs", 1023), ("allA", 5), ("lzmix", 40000), ("text", 0)):
            src = datagen.gen_bytes(kind, n, 17)
            rc, used, out, crc = s.compress(src, 1, crc0=0)
            erc, eused, exp, ecrc = O.sw_compress(fmt, src, hw, 1, cap=n * 9 // 8 + 65536)
            assert rc == A.QZ_OK and used == n, (fmt, hw, kind, n, rc)
            assert out == exp, (fmt, hw, kind, n, len(out), len(exp))
            assert crc == ecrc, (fmt, hw, kind, n)
            if fmt in ("GZIP", "GZIP_EXT") and n:
                assert zlib.decompress(out, 31) == src
            rc, cused, back = s.decompress(out, n + 16)
            if n:
                assert rc == A.QZ_OK and back == src and cused == len(out), (fmt, hw, kind, n, rc)
        s.close()


def test_zlib_format_sessions_match_sw_path():
    """qzSetupSessionDeflateExt(zlib_format = 1): RFC 1950 header, the same chunked deflate body, Adler-32 trailer
    (src/qatzip_sw.c:147 with windowBits 15) - bit-identical to the oracle, readable by zlib.decompress"""
    for hw in (65536, 16384):
        s = A.Session(hw_buff_sz=hw, zlib_format=True)
        assert s.rc_setup == A.QZ_OK
        for kind, n in (("silesia", 300123), ("text", 65536), ("rand", 70000), ("allA", 200000), ("text", 0), ("runs", 999)):
            src = datagen.gen_bytes(kind, n, 23)
            rc, used, out, crc = s.compress(src, 1, crc0=0)
            erc, eused, exp, ecrc = O.sw_compress("ZLIB", src, hw, 1, cap=n * 9 // 8 + 65536)
            assert rc == A.QZ_OK and used == n and out == exp and crc == ecrc, (hw, kind, n, rc)
            assert zlib.decompress(out) == src
            rc, cused, back = s.decompress(out, n + 16)
            if n:
                assert rc == A.QZ_OK and back == src and cused == len(out)
        # two calls, one stream (last = 0 then 1), and a damaged trailer
        src = datagen.gen_bytes("silesia", 200000, 5)
        rc1, u1, o1, _ = s.compress(src[:131072], 0)
        rc2, u2, o2, _ = s.compress(src[131072:], 1)
        assert rc1 == A.QZ_OK and rc2 == A.QZ_OK and zlib.decompress(o1 + o2) == src
        bad = bytearray(o1 + o2); bad[-1] ^= 1
        rc, _, _ = s.decompress(bytes(bad), len(src) + 16)
        assert rc == A.QZ_DATA_ERROR
        foreign = zlib.compress(src, 6)                          # another producer's zlib stream
        rc, cused, back = s.decompress(foreign, len(src) + 16)
        assert rc == A.QZ_OK and back == src and cused == len(foreign)
        s.close()


def _hw_path_stream(src, hw):
    """what the reference's hardware path writes: one gzip-ext member per hw_buff_sz chunk, both sizes in the header
    (src/qatzip_gzip.c:86-143), here with zlib as the per-chunk deflate engine"""
    out = []
    for off in range(0, len(src), hw):
        chunk = src[off:off + hw]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        hdr = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0]) + b"QZ" + (8).to_bytes(2, "little") + \
            len(chunk).to_bytes(4, "little") + len(body).to_bytes(4, "little")
        out.append(hdr + body + (zlib.crc32(chunk) & 0xffffffff).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
    return out


def test_hardware_path_streams_decode_as_one_batch():
    """interop with data compressed on real QAT boxes: thousands of sized members per call, decoded by one launch"""
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    src = datagen.gen_bytes("silesia", 20 << 20, 77) + datagen.gen_bytes("rand", 100000, 1) + datagen.gen_bytes("text", 12345, 2)
    members = _hw_path_stream(src, 65536)
    comp = b"".join(members)
    crc = C.c_ulong(0)
    sl, dl = C.c_uint(len(comp)), C.c_uint(len(src) + 64)
    dst = C.create_string_buffer(len(src) + 64)
    rc = s.L.qzDecompressCrc(C.byref(s.s), comp, C.byref(sl), dst, C.byref(dl), C.byref(crc))
    assert rc == A.QZ_OK and sl.value == len(comp) and dst.raw[:dl.value] == src
    assert crc.value == (zlib.crc32(src) & 0xffffffff)
    # destination for 100 members only: whole members, QZ_BUF_ERROR, resumable
    rc, used, out = s.decompress(comp, 100 * 65536 + 1000)
    assert rc == A.QZ_BUF_ERROR and out == src[:100 * 65536] and used == sum(len(m) for m in members[:100])
    rc, used2, out2 = s.decompress(comp[used:], len(src))
    assert rc == A.QZ_OK and out + out2 == src
    # a damaged member in the middle: the members before it come out, the next call reports the error
    k = 150
    bad = bytearray(comp); pos = sum(len(m) for m in members[:k]) + 24 + 10; bad[pos] ^= 0x5a
    rc, used, out = s.decompress(bytes(bad), len(src) + 64)
    assert rc == A.QZ_OK and out == src[:k * 65536] and used == sum(len(m) for m in members[:k])
    rc, _, _ = s.decompress(bytes(bad[used:]), len(src))
    assert rc == A.QZ_DATA_ERROR
    # stream cut inside a member
    cut = sum(len(m) for m in members[:40]) + 5000
    rc, used, out = s.decompress(comp[:cut], len(src))
    assert rc == A.QZ_OK and used == sum(len(m) for m in members[:40]) and out == src[:40 * 65536]
    s.close()


@pytest.mark.parametrize("fmt", ["GZIP_EXT", "GZIP", "4B"])
def test_hardware_path_framing_on_the_compress_side(fmt):
    """qzamd_set_hw_framing(sess, 1): one complete member per hw_buff_sz chunk, as the reference's engine retires them
    (src/qatzip.c:1691-1718; headers src/qatzip_gzip.c:86-143: XFL 0, OS 255, both sizes in the gzip-ext field) - the
    stream layout of test_hardware_path_streams_decode_as_one_batch, with this library's own deflate inside"""
    L = A.lib()
    L.qzamd_set_hw_framing.argtypes = [C.c_void_p, C.c_int]
    for hw in (65536, 16384):
        s = A.Session(FMT[fmt], hw)
        assert L.qzamd_set_hw_framing(C.byref(s.s), 1) == A.QZ_OK
        for kind, n in (("silesia", 5 * hw + 777), ("rand", 2 * hw), ("text", hw), ("runs", 1024), ("allA", 3 * hw + 1)):
            src = datagen.gen_bytes(kind, n, 31)
            rc, used, out, crc = s.compress(src, 1, crc0=0)
            assert rc == A.QZ_OK and used == n, (fmt, hw, kind, rc)
            exp = b""
            for off in range(0, n, hw):
                chunk = src[off:off + hw]
                body = O.sw_compress("RAW", chunk, hw, 1, last=1)[2]          # a closed deflate stream per chunk
                if fmt == "GZIP_EXT":
                    hdr = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0]) + b"QZ" + (8).to_bytes(2, "little") + \
                        len(chunk).to_bytes(4, "little") + len(body).to_bytes(4, "little")
                elif fmt == "GZIP":
                    hdr = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255])
                else:
                    hdr = len(body).to_bytes(4, "little")
                ftr = b"" if fmt == "4B" else (zlib.crc32(chunk) & 0xffffffff).to_bytes(4, "little") + len(chunk).to_bytes(4, "little")
                exp += hdr + body + ftr
            assert out == exp, (fmt, hw, kind, n, len(out), len(exp))
            assert crc == (zlib.crc32(src) & 0xffffffff)
            if fmt != "4B":
                assert b"".join(zlib.decompress(m, 31) for m in _split_members(out, fmt)) == src
            rc, cused, back = s.decompress(out, n + 64)
            assert rc == A.QZ_OK and back == src and cused == len(out), (fmt, hw, kind, rc)
        # calls below input_sz_thrshold (1024 by default) - the empty call too - go to the reference's software path even
        # on a QAT box (src/qatzip.c:1934-1947), so they keep its framing
        for n in (0, 1, 1000, 1023):
            small = datagen.gen_bytes("runs", n, 32)
            rc, used, out, _ = s.compress(small, 1)
            assert rc == A.QZ_OK and used == n and out == O.sw_compress(fmt, small, hw, 1, cap=n + 4096)[2], (fmt, hw, n)
        # a destination for two members only: whole members, QZ_BUF_ERROR, the caller resumes behind them
        src = datagen.gen_bytes("text", 4 * hw, 5)
        full = s.compress(src, 1)[2]
        two = len(_split_members(full, fmt)[0]) + len(_split_members(full, fmt)[1]) if fmt != "4B" else None
        if two:
            rc, used, out, _ = s.compress(src, 1, cap=two + 10)
            assert rc == A.QZ_BUF_ERROR and used == 2 * hw and out == full[:two]
        assert L.qzamd_set_hw_framing(C.byref(s.s), 0) == A.QZ_OK
        assert s.compress(src, 1)[2] == O.sw_compress(fmt, src, hw, 1, cap=len(src) * 9 // 8 + 65536)[2]
        s.close()


def _split_members(buf, fmt):
    """cut a sequence of gzip members apart with zlib's own reader"""
    out, pos = [], 0
    while pos < len(buf):
        d = zlib.decompressobj(31)
        d.decompress(buf[pos:])
        end = len(buf) - len(d.unused_data)
        out.append(buf[pos:end]); pos = end
    return out


def test_async_compress2_decompress2():
    """qzCompress2 / qzDecompress2 (src/qatzip.c:4112-4196): callback == NULL is the synchronous call; with a callback
    the request is queued, QZ_OK comes back at once, and a library thread retires it and reports through QzResult_T"""
    import threading
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    L = s.L
    srcs = [datagen.gen_bytes(k, n, 40 + i) for i, (k, n) in enumerate(
        (("silesia", 300000), ("text", 65536), ("rand", 100000), ("runs", 5000), ("lzmix", 66000), ("records", 1 << 20)))]
    bufs_in = [C.create_string_buffer(x, len(x)) for x in srcs]
    bufs_out = [C.create_string_buffer(len(x) * 9 // 8 + 4096) for x in srcs]
    results = [A.QzResult() for _ in srcs]
    done, order = threading.Event(), []

    def on_done(res):
        order.append(res.contents.cb_tag)
        if len(order) == len(srcs):
            done.set()
        return 0
    cb = A.QzAsyncCallback(on_done)
    for i, x in enumerate(srcs):
        results[i].cb_tag = i + 1; results[i].src_len = len(x); results[i].dest_len = len(bufs_out[i])
        assert L.qzCompress2(C.byref(s.s), bufs_in[i], bufs_out[i], cb, C.byref(results[i])) == A.QZ_OK
    assert done.wait(120)
    assert order == list(range(1, len(srcs) + 1))                # retired in submission order
    comp = []
    for i, x in enumerate(srcs):
        r = results[i]
        assert r.status == A.QZ_OK and r.src_len == len(x)
        comp.append(bufs_out[i].raw[:r.dest_len])
        assert comp[i] == O.sw_compress("GZIP_EXT", x, 65536, 1, cap=len(x) * 9 // 8 + 65536)[2]
    # decompress: one asynchronous request, one synchronous (callback NULL)
    done.clear(); order.clear()
    back = C.create_string_buffer(len(srcs[0]) + 64)
    r = A.QzResult(); r.cb_tag = 99; r.src_len = len(comp[0]); r.dest_len = len(back)
    cin = C.create_string_buffer(comp[0], len(comp[0]))

    def on_one(res):
        done.set()
        return 0
    cb1 = A.QzAsyncCallback(on_one)
    assert L.qzDecompress2(C.byref(s.s), cin, back, cb1, C.byref(r)) == A.QZ_OK
    assert done.wait(120) and r.status == A.QZ_OK and back.raw[:r.dest_len] == srcs[0] and r.src_len == len(comp[0])
    r2 = A.QzResult(); r2.src_len = len(comp[1]); r2.dest_len = len(back)
    cin2 = C.create_string_buffer(comp[1], len(comp[1]))
    assert L.qzDecompress2(C.byref(s.s), cin2, back, None, C.byref(r2)) == A.QZ_OK
    assert r2.status == A.QZ_OK and back.raw[:r2.dest_len] == srcs[1]
    assert L.qzCompress2(C.byref(s.s), cin2, back, None, None) == A.QZ_PARAMS
    s.close()                                                    # teardown waits for the queue of this session


def test_async_queue_coalesces_small_requests():
    """Many small qzCompress2 requests in flight (the reference's async test, test/main.c:5527-6374, keeps the ring
    full): whatever is waiting goes to the GPU as one launch, and every request still gets exactly the member a call of
    its own would have written - sizes 0 .. 3 chunks, two sessions sharing launches, a third with another format in
    between, one destination too small, one request with the CRC out-field."""
    import threading
    import time
    sa, sb = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536), A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    sr = A.Session(A.QZ_DEFLATE_RAW, 16384, comp_lvl=4)
    L = sa.L
    L.qzamd_async_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    l0, r0 = C.c_uint64(), C.c_uint64()
    L.qzamd_async_stats(C.byref(l0), C.byref(r0))
    kinds = ("silesia", "text", "rand", "runs", "records", "lzmix")
    sizes = [65536, 1000, 0, 65537, 30000, 131072, 1, 200000, 65535, 4096]
    jobs = []                                                   # (session, fmt, hw, lvl, src)
    for i in range(240):
        sess, fmt, hw, lvl = ((sa, "GZIP_EXT", 65536, 1), (sb, "GZIP_EXT", 65536, 1), (sr, "RAW", 16384, 4))[0 if i % 7 < 4 else 1 if i % 7 < 6 else 2]
        n = sizes[i % len(sizes)]
        if kinds[i % 6] == "lzmix":
            n = min(n, 66000)
        jobs.append((sess, fmt, hw, lvl, datagen.gen_bytes(kinds[i % 6], n, 300 + i)))
    bufs_in = [C.create_string_buffer(j[4], max(1, len(j[4]))) for j in jobs]
    bufs_out = [C.create_string_buffer(len(j[4]) * 9 // 8 + 4096) for j in jobs]
    results = [A.QzResult() for _ in jobs]
    small = 17                                                  # this one gets a destination that cannot hold its member
    done, order = threading.Event(), []

    def on_done(res):
        order.append(res.contents.cb_tag)
        if len(order) == len(jobs):
            done.set()
        return 0
    cb = A.QzAsyncCallback(on_done)
    t0 = time.time()
    for i, j in enumerate(jobs):
        results[i].cb_tag = i + 1; results[i].src_len = len(j[4]); results[i].dest_len = 64 if i == small else len(bufs_out[i])
        assert L.qzCompress2(C.byref(j[0].s), bufs_in[i], bufs_out[i], cb, C.byref(results[i])) == A.QZ_OK
    assert done.wait(300)
    dt = time.time() - t0
    assert order == list(range(1, len(jobs) + 1))                # still retired in submission order
    for i, j in enumerate(jobs):
        r = results[i]
        if i == small:
            assert r.status in (A.QZ_BUF_ERROR, A.QZ_FAIL) or r.dest_len <= 64
            continue
        exp = O.sw_compress(j[1], j[4], j[2], j[3], cap=len(j[4]) * 9 // 8 + 65536)[2]
        assert r.status == A.QZ_OK and r.src_len == len(j[4]) and bufs_out[i].raw[:r.dest_len] == exp, (i, j[1], len(j[4]), r.status)
    l1, r1 = C.c_uint64(), C.c_uint64()
    L.qzamd_async_stats(C.byref(l1), C.byref(r1))
    nl, nr = l1.value - l0.value, r1.value - r0.value
    print("async: %d requests in %.3f s, %d coalesced launches carried %d of them" % (len(jobs), dt, nl, nr))
    assert nr >= len(jobs) // 2 and nl < nr                      # most of them shared a launch
    assert sa.s.total_in + sb.s.total_in + sr.s.total_in >= sum(len(j[4]) for k, j in enumerate(jobs) if k != small)
    # and back: every gzip-ext member as its own qzDecompress2 request, all in flight; one of them damaged
    back_jobs = [i for i, j in enumerate(jobs) if j[1] == "GZIP_EXT" and i != small]
    cins = [C.create_string_buffer(bufs_out[i].raw[:results[i].dest_len], results[i].dest_len) for i in back_jobs]
    bad = 5
    raw = bytearray(cins[bad].raw); raw[len(raw) // 2] ^= 0x10
    cins[bad] = C.create_string_buffer(bytes(raw), len(raw))
    douts = [C.create_string_buffer(len(jobs[i][4]) + 32) for i in back_jobs]
    dres = [A.QzResult() for _ in back_jobs]
    done.clear(); order.clear()

    def on_back(res):
        order.append(res.contents.cb_tag)
        if len(order) == len(back_jobs):
            done.set()
        return 0
    cb2 = A.QzAsyncCallback(on_back)
    for k, i in enumerate(back_jobs):
        dres[k].cb_tag = k + 1; dres[k].src_len = len(cins[k]); dres[k].dest_len = len(douts[k])
        assert L.qzDecompress2(C.byref(jobs[i][0].s), cins[k], douts[k], cb2, C.byref(dres[k])) == A.QZ_OK
    assert done.wait(300) and order == list(range(1, len(back_jobs) + 1))
    for k, i in enumerate(back_jobs):
        if k == bad:
            assert dres[k].status == A.QZ_DATA_ERROR
        elif len(jobs[i][4]):
            assert dres[k].status == A.QZ_OK and douts[k].raw[:dres[k].dest_len] == jobs[i][4] and dres[k].src_len == len(cins[k]), (k, i)
    l2, r2 = C.c_uint64(), C.c_uint64()
    L.qzamd_async_stats(C.byref(l2), C.byref(r2))
    print("async back: %d requests, %d coalesced launches carried %d" % (len(back_jobs), l2.value - l1.value, r2.value - r1.value))
    assert r2.value - r1.value >= len(back_jobs) // 2
    sa.close(); sb.close(); sr.close()


@pytest.mark.parametrize("lvl", [2, 3])
def test_async_coalesced_levels_2_3_regrow_lane_scratch(lvl):
    """Coalesced qzCompress2 launches at the greedy levels above 1 go through the one-chunk-per-lane path, whose scratch
    grows with the batch: a fresh session, a small batch, then a larger one that regrows it (the per-slot descriptor
    buffer of the launch must survive the regrow - ADVICE r1)."""
    import threading
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536, comp_lvl=lvl)
    L = s.L
    for rnd, (count, n) in enumerate(((6, 20000), (40, 150000))):
        srcs = [datagen.gen_bytes(("silesia", "text", "runs", "records")[i % 4], n + 37 * i, 900 + 50 * rnd + i) for i in range(count)]
        bin_ = [C.create_string_buffer(x, len(x)) for x in srcs]
        bout = [C.create_string_buffer(len(x) * 9 // 8 + 4096) for x in srcs]
        res = [A.QzResult() for _ in srcs]
        done, got = threading.Event(), []

        def on_done(r):
            got.append(r.contents.cb_tag)
            if len(got) == len(srcs):
                done.set()
            return 0
        cb = A.QzAsyncCallback(on_done)
        for i, x in enumerate(srcs):
            res[i].cb_tag = i + 1; res[i].src_len = len(x); res[i].dest_len = len(bout[i])
            assert L.qzCompress2(C.byref(s.s), bin_[i], bout[i], cb, C.byref(res[i])) == A.QZ_OK
        assert done.wait(300)
        for i, x in enumerate(srcs):
            exp = O.sw_compress("GZIP_EXT", x, 65536, lvl, cap=len(x) * 9 // 8 + 65536)[2]
            assert res[i].status == A.QZ_OK and bout[i].raw[:res[i].dest_len] == exp, (lvl, rnd, i)
    s.close()


def test_async_teardown_from_callback_does_not_deadlock():
    """a completion callback may tear down a session of its own batch: callbacks run after the whole batch
    has left the running list"""
    import threading
    sa, sb = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536), A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    L = sa.L
    srcs = [datagen.gen_bytes("text", 30000 + i, 70 + i) for i in range(8)]
    bin_ = [C.create_string_buffer(x, len(x)) for x in srcs]
    bout = [C.create_string_buffer(len(x) * 9 // 8 + 4096) for x in srcs]
    res = [A.QzResult() for _ in srcs]
    done, n_cb, torn = threading.Event(), [0], [False]

    def on_done(r):
        n_cb[0] += 1
        if not torn[0]:
            torn[0] = True
            assert L.qzTeardownSession(C.byref(sb.s)) == A.QZ_OK      # sb's requests sit in this very batch
        if n_cb[0] == len(srcs):
            done.set()
        return 0
    cb = A.QzAsyncCallback(on_done)
    for i, x in enumerate(srcs):
        res[i].cb_tag = i + 1; res[i].src_len = len(x); res[i].dest_len = len(bout[i])
        assert L.qzCompress2(C.byref((sa if i % 2 == 0 else sb).s), bin_[i], bout[i], cb, C.byref(res[i])) == A.QZ_OK
    assert done.wait(120), "consumer thread deadlocked in a callback"
    for i, x in enumerate(srcs):
        if res[i].status == A.QZ_OK:
            assert zlib.decompress(bout[i].raw[:res[i].dest_len], 31) == x
    assert sum(1 for r in res if r.status == A.QZ_OK) >= len(srcs) // 2
    sa.close()


def test_synchronous_calls_of_many_threads_share_launches():
    """qzCompress / qzDecompress of small requests issued by several threads at once (the reference's perf harness,
    test/main.c:2175-2299, one call per block per thread): the calls wait in the same queue as qzCompress2 requests and
    whatever is waiting goes to the GPU as one launch; every caller still gets exactly the member its own call would have
    written, its own lengths, return code and CRC."""
    import threading
    L = A.lib()
    L.qzamd_async_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    l0, r0 = C.c_uint64(), C.c_uint64()
    L.qzamd_async_stats(C.byref(l0), C.byref(r0))
    NT, PER = 8, 12
    errs, start = [], threading.Barrier(NT)

    def body(t):
        try:
            s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
            s.compress(b"warm up the session", 1)
            start.wait()
            for i in range(PER):
                kind = ("silesia", "text", "rand", "runs")[(t + i) % 4]
                n = (65536, 1000, 200000, 65537, 30000)[(t * 3 + i) % 5]
                src = datagen.gen_bytes(kind, n, 1000 + 50 * t + i)
                rc, used, out, crc = s.compress(src, 1, crc0=0)
                exp = O.sw_compress("GZIP_EXT", src, 65536, 1, cap=n * 9 // 8 + 65536)
                assert rc == A.QZ_OK and used == n and out == exp[2] and crc == exp[3], (t, i, kind, n, rc)
                rc, cused, back, dcrc = s.decompress(out, n + 16, want_crc=True)
                assert rc == A.QZ_OK and back == src and cused == len(out) and dcrc == (zlib.crc32(src) & 0xffffffff), (t, i, rc)
                if i == 5:                                          # a destination that is too small: this caller's error only
                    rc, used, out2, _ = s.compress(src, 1, cap=40)
                    assert rc in (A.QZ_BUF_ERROR, A.QZ_FAIL) and used == 0
            s.close()
        except Exception as e:   # noqa: BLE001
            errs.append((t, repr(e)))
    th = [threading.Thread(target=body, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    for x in th:
        x.join(600)
    assert not errs, errs
    l1, r1 = C.c_uint64(), C.c_uint64()
    L.qzamd_async_stats(C.byref(l1), C.byref(r1))
    print("sync calls of %d threads: %d coalesced launches carried %d requests" % (NT, l1.value - l0.value, r1.value - r0.value))
    assert r1.value - r0.value >= NT                         # calls did share launches


def test_large_calls_of_several_threads_share_the_device_tables():
    """one session per thread, every thread making calls far above the coalescing limit at the same time: the LZ77
    candidate tables belong to the device, a context borrows them for the duration of a call"""
    import threading
    errs, start = [], threading.Barrier(3)

    def body(t):
        try:
            s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
            src = datagen.gen_bytes(("silesia", "text", "records")[t], (24 << 20) + 4099 * t, 70 + t)
            exp = O.sw_compress("GZIP_EXT", src, 65536, 1, cap=len(src) * 9 // 8 + 65536)[2]
            start.wait()
            for _ in range(3):
                rc, used, out, _ = s.compress(src, 1)
                assert rc == A.QZ_OK and used == len(src) and out == exp, (t, rc)
                rc, cused, back = s.decompress(out, len(src) + 64)
                assert rc == A.QZ_OK and back == src, (t, rc)
            s.close()
        except Exception as e:   # noqa: BLE001
            errs.append((t, repr(e)[:300]))
    th = [threading.Thread(target=body, args=(t,)) for t in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join(900)
    assert not errs, errs


def test_crc_known_answer_like_reference_test():
    # test/main.c:4283-4337: qzCompressCrc's crc == zlib crc32(src) for 64 KB and 1023 B
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    for n in (65536, 1023):
        src = datagen.gen_bytes("runs", n, 3)
        rc, _, _, crc = s.compress(src, 1, crc0=0)
        assert rc == 0 and crc == (zlib.crc32(src) & 0xffffffff)
    s.close()


def test_stream_continues_across_calls_like_sw_path():
    # last=0 then last=1: one member, sizes in the gzip-ext header stay 0
    src = datagen.gen_bytes("text", 150000, 4)
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    rc1, u1, o1, _ = s.compress(src[:100000], 0)
    rc2, u2, o2, _ = s.compress(src[100000:], 1)
    assert rc1 == 0 and rc2 == 0 and u1 == 100000 and u2 == 50000
    out = o1 + o2
    assert o1[-4:] == b"\x00\x00\xff\xff" and out[16:24] == b"\0" * 8
    assert zlib.decompress(out, 31) == src
    # the first call equals the oracle's view of an open-ended stream
    assert o1 == O.sw_compress("GZIP_EXT", src[:100000], 65536, 1, last=0)[2]
    s.close()


def test_multi_member_decompress_and_buf_error():
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    parts = [datagen.gen_bytes("silesia", n, 30 + i) for i, n in enumerate((65536, 1000, 140000))]
    comp = b"".join(s.compress(p, 1)[2] for p in parts)
    rc, used, back = s.decompress(comp, sum(map(len, parts)) + 10)
    assert rc == A.QZ_OK and used == len(comp) and back == b"".join(parts)
    # a multi-chunk member FIRST (qzCompress fills the size fields of the header for those too, so it looks like the
    # hardware path's one-chunk members until it is decoded), then small ones: the batch decode steps aside
    parts2 = [datagen.gen_bytes("silesia", n, 40 + i) for i, n in enumerate((300000, 65536, 10, 140000))]
    comp2 = b"".join(s.compress(p, 1)[2] for p in parts2)
    rc, used, back = s.decompress(comp2, sum(map(len, parts2)) + 16)
    assert rc == A.QZ_OK and used == len(comp2) and back == b"".join(parts2)
    # destination that only holds the first member: progress is reported, caller loops (utils/qzip.c:217-227)
    rc, used, back = s.decompress(comp, 65536 + 500)
    assert rc in (A.QZ_OK, A.QZ_BUF_ERROR) and back == parts[0] and 0 < used < len(comp)
    # compress into a tiny destination: QZ_BUF_ERROR, nothing consumed (HW-path contract, test/main.c:4264-4270)
    rc, used, out, _ = s.compress(parts[2], 1, cap=1024)
    assert rc == A.QZ_BUF_ERROR and used == 0 and out == b""
    # partial progress: room for some chunks only
    s2 = A.Session(A.QZ_DEFLATE_RAW, 65536)
    rnd = datagen.gen_bytes("rand", 200000, 2)
    rc, used, out, _ = s2.compress(rnd, 1, cap=140000)
    assert rc == A.QZ_BUF_ERROR and used == 131072 and zlib.decompressobj(-15).decompress(out) == rnd[:131072]
    s.close(); s2.close()


def test_mixed_format_members_decode_in_one_call():
    """test/main.c mode 5 (:892-1111): blocks compressed alternately as gzip-ext ("QZ") and plain gzip, concatenated,
    then one decompress loop over the lot; plus the L9-software-stream case (:4339-4409) and members written by
    another gzip implementation in between."""
    import gzip
    se, sg = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536), A.Session(A.QZ_DEFLATE_GZIP, 65536)
    s9 = A.Session(A.QZ_DEFLATE_GZIP, 65536, comp_lvl=9)
    parts, comp = [], b""
    for i, (who, n) in enumerate((("QZ", 200000), ("GZIP", 70000), ("QZ", 65536), ("L9", 150000), ("PY", 90000), ("GZIP", 1), ("QZ", 300000))):
        src = datagen.gen_bytes(("silesia", "text", "records")[i % 3], n, 60 + i)
        parts.append(src)
        if who == "PY":
            comp += gzip.compress(src, 6)
        else:
            rc, used, out, _ = {"QZ": se, "GZIP": sg, "L9": s9}[who].compress(src, 1)
            assert rc == A.QZ_OK and used == n
            comp += out
    want = b"".join(parts)
    for s in (se, sg):                                   # either session kind reads the mix
        got, pos = b"", 0
        while pos < len(comp):                           # the caller's loop (utils/qzip.c:217-227)
            rc, used, back = s.decompress(comp[pos:], len(want) + 16)
            assert rc == A.QZ_OK and used > 0, (rc, pos)
            got += back; pos += used
        assert got == want
    se.close(); sg.close(); s9.close()


def test_member_larger_than_the_destination_comes_out_in_pieces():
    """SURVEY 8b "decompress into 1 KB dest": the software path hands out what fits, returns QZ_OK and resumes on the
    next call.  Same contract here for a destination smaller than the member: the caller's loop (advance by consumed,
    collect produced) gets every byte, the running CRC is right, and the stream after the member decodes normally."""
    import gzip
    src = datagen.gen_bytes("silesia", 300_000, 71)
    tail_src = datagen.gen_bytes("text", 50_000, 72)
    for fmt, cap in (("GZIP_EXT", 1000), ("GZIP", 70_000), ("RAW", 4096)):
        s = A.Session(FMT[fmt], 65536)
        comp = s.compress(src, 1)[2] + s.compress(tail_src, 1)[2]
        got, pos, crc, calls = b"", 0, 0, 0
        while pos < len(comp):
            rc, used, back, crc = s.decompress(comp[pos:], cap if len(got) < len(src) else 60_000, crc0=crc, want_crc=True)
            assert rc == A.QZ_OK and (used or back), (fmt, rc, pos)
            got += back; pos += used; calls += 1
            assert calls < 1000
        assert got == src + tail_src and pos == len(comp) and calls > 4, (fmt, calls)
        assert crc == zlib.crc32(src + tail_src) & 0xffffffff, fmt
        s.close()
    # a member written by another gzip, far larger than the destination
    big = datagen.gen_bytes("records", 2_000_000, 73)
    comp = gzip.compress(big, 6)
    s = A.Session(A.QZ_DEFLATE_GZIP, 65536)
    got, pos = b"", 0
    while pos < len(comp):
        rc, used, back = s.decompress(comp[pos:], 65536)
        assert rc == A.QZ_OK and (used or back)
        got += back; pos += used
    assert got == big
    s.close()


@pytest.mark.parametrize("fmt", ["GZIP_EXT", "GZIP", "ZLIB"])
def test_stop_decompression_on_stream_end(fmt):
    """test/main.c:1305-1702 (modes 27, 30, 31): 512 KB compressed as eight separate 64 KB streams; with
    stop_decompression_stream_end set, one qzDecompress call over all of them decodes exactly the first stream and
    qzGetDeflateEndOfStream reports 1; a cut-off stream is an error and reports 0; without the flag every stream is
    decoded."""
    A.lib().qzGetDeflateEndOfStream.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte)]
    src = datagen.gen_bytes("silesia", 512 * 1024, 88)
    mk = (lambda **kw: A.Session(hw_buff_sz=65536, zlib_format=True, **kw)) if fmt == "ZLIB" else (lambda **kw: A.Session(FMT[fmt], 65536, **kw))
    s = mk(stop_at_stream_end=True)
    assert s.rc_setup == A.QZ_OK
    members = [s.compress(src[o:o + 65536], 1)[2] for o in range(0, len(src), 65536)]
    comp = b"".join(members)
    rc, used, back = s.decompress(comp, len(src))
    assert rc == A.QZ_OK and back == src[:65536] and used == len(members[0]), (fmt, rc, used, len(back))
    assert s.end_of_stream() == (A.QZ_OK, 1)
    rc, used, back = s.decompress(comp[:1100], len(src))
    assert rc != A.QZ_OK and back == b""
    assert s.end_of_stream() == (A.QZ_OK, 0)
    # the caller's loop gets the rest stream by stream
    pos, got = 0, b""
    while pos < len(comp):
        rc, used, back = s.decompress(comp[pos:], len(src))
        assert rc == A.QZ_OK and used > 0 and s.end_of_stream()[1] == 1
        got += back; pos += used
    assert got == src
    s.close()
    s2 = mk()
    rc, used, back = s2.decompress(comp, len(src))
    assert rc == A.QZ_OK and back == src and used == len(comp)
    s2.close()


def test_corrupt_input_is_a_data_error():
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    src = datagen.gen_bytes("text", 90000, 6)
    comp = s.compress(src, 1)[2]
    bad = bytearray(comp); bad[0] ^= 0xff
    rc, used, back = s.decompress(bytes(bad), len(src))
    assert rc == A.QZ_DATA_ERROR and used == 0 and back == b""
    bad = bytearray(comp); bad[len(comp) // 2] ^= 0x10
    rc, used, back = s.decompress(bytes(bad), len(src))
    assert rc == A.QZ_DATA_ERROR and used == 0 and back == b""
    s.close()


def test_every_comp_lvl_matches_sw_path():
    """comp_lvl 1-9 through qzCompress: the deflate body of that zlib level and the level-dependent header bytes (gzip
    XFL 4 / 0 / 2, zlib FLEVEL) exactly as the software path writes them; levels zlib does not have fail loudly."""
    src = datagen.gen_bytes("silesia", 300123, 41)
    for lvl in range(1, 10):
        for fmt in ("GZIP_EXT", "GZIP", "ZLIB"):
            s = A.Session(hw_buff_sz=65536, comp_lvl=lvl, zlib_format=True) if fmt == "ZLIB" else A.Session(FMT[fmt], 65536, comp_lvl=lvl)
            assert s.rc_setup == A.QZ_OK
            rc, used, out, crc = s.compress(src, 1, crc0=0)
            erc, eused, exp, ecrc = O.sw_compress(fmt, src, 65536, lvl, cap=len(src) * 9 // 8 + 65536)
            assert rc == A.QZ_OK and used == len(src) and out == exp and crc == ecrc, (lvl, fmt, rc)
            assert zlib.decompress(out, 15 if fmt == "ZLIB" else 31) == src
            rc, cused, back = s.decompress(out, len(src) + 16)
            assert rc == A.QZ_OK and back == src and cused == len(out)
            s.close()
    s = A.Session(hw_buff_sz=65536, comp_lvl=12, zlib_format=True)      # the new API admits 10-12 (QAT gen 3); zlib has no such levels
    if s.rc_setup == A.QZ_OK:
        rc, used, out, _ = s.compress(b"x" * 1000, 1)
        assert rc == A.QZ_NOT_SUPPORTED and used == 0 and out == b""
    s.close()


def test_one_session_per_thread_runs_concurrently():
    """The reference's threading contract (include/qatzip.h:122-148, test/main.c -t): the API is thread-safe with one
    session per thread.  Four threads, each with its own session and format, compress and decompress at the same
    time (ctypes drops the GIL inside the calls); every result is what the software path writes."""
    import threading
    jobs = [("GZIP_EXT", 65536, "silesia", 3_000_000), ("GZIP", 16384, "text", 1_500_000), ("RAW", 131072, "records", 2_000_000),
            ("4B", 65536, "lzmix", 140_000)]
    errs = []
    O.sw_compress("RAW", b"warm the checker's static tables on this thread", 65536, 1)

    def worker(fmt, hw, kind, n, seed):
        try:
            s = A.Session(FMT[fmt], hw)
            assert s.rc_setup == A.QZ_OK
            for it in range(3):
                src = datagen.gen_bytes(kind, n - it * 1000, seed + it)
                rc, used, out, crc = s.compress(src, 1, crc0=0)
                assert rc == A.QZ_OK and used == len(src), (fmt, rc)
                assert out == O.sw_compress(fmt, src, hw, 1, cap=len(src) * 9 // 8 + 65536)[2], (fmt, it)
                rc, cused, back = s.decompress(out, len(src) + 16)
                assert rc == A.QZ_OK and back == src and cused == len(out), (fmt, it, rc)
            s.close()
        except BaseException as e:      # noqa: BLE001 - reported on the main thread
            errs.append((fmt, repr(e)))

    ts = [threading.Thread(target=worker, args=(*j, 50 + i)) for i, j in enumerate(jobs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errs, errs


def test_stream_api_one_member_many_slices():
    # qzCompressStream fed in hw_buff_sz/4 slices (test/main.c:2548): one gzip-ext member, correct trailer
    L = A.lib()
    src = datagen.gen_bytes("silesia", 250000, 8)
    for sb in (65536, 262144):
        sess = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536, strm_buff_sz=sb)
        strm = A.QzStream()
        out = b""
        obuf = C.create_string_buffer(1 << 20)
        pos = 0
        while True:
            k = min(16384, len(src) - pos)
            piece = src[pos:pos + k]
            ibuf = C.create_string_buffer(piece, max(k, 1))
            strm.in_ = C.cast(ibuf, C.c_void_p); strm.in_sz = k
            strm.out = C.cast(obuf, C.c_void_p); strm.out_sz = len(obuf)
            last = 1 if pos + k == len(src) else 0
            rc = L.qzCompressStream(C.byref(sess.s), C.byref(strm), last)
            assert rc == A.QZ_OK
            pos += strm.in_sz
            out += obuf.raw[:strm.out_sz]
            if last and strm.in_sz == k and strm.pending_out == 0 and strm.pending_in == 0:
                break
        assert zlib.decompress(out, 31) == src
        assert out[:4] == b"\x1f\x8b\x08\x04"
        if sb < len(src):       # stream opened by a non-final slab: size fields are never patched (src/qatzip_sw.c:166,238)
            assert out[16:24] == b"\0" * 8
        else:                   # the only slab is also the last: same bytes as a single qzCompress(last=1)
            assert out == O.sw_compress("GZIP_EXT", src, 65536, 1)[2]
        L.qzEndStream(C.byref(sess.s), C.byref(strm))
        sess.close()


@pytest.mark.parametrize("out_room", [1 << 20, 50_000])
def test_decompress_stream_fed_in_slices(out_room):
    """qzDecompressStream the way the reference's stream tests drive it (test/main.c:2506-2847): compressed data arrives
    in slices that cut members anywhere, output is collected from a fixed buffer - here also one smaller than the
    members, so they come out in pieces.  Members of three producers; running CRC; damaged input; LZ4 sessions refused."""
    import gzip
    L = A.lib()
    parts = [datagen.gen_bytes("silesia", 300_000, 91), datagen.gen_bytes("text", 65_536, 92), datagen.gen_bytes("runs", 1000, 93),
             datagen.gen_bytes("records", 150_000, 94)]
    cs = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    comp = b"".join(cs.compress(x, 1)[2] for x in parts[:3]) + gzip.compress(parts[3], 6)
    want = b"".join(parts)
    cs.close()

    def run(data):
        sess = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
        strm = A.QzStream()
        obuf = C.create_string_buffer(out_room)
        got, pos, rc, calls = b"", 0, A.QZ_OK, 0
        while True:
            k = min(10_000, len(data) - pos)
            ibuf = C.create_string_buffer(data[pos:pos + k], max(k, 1))
            strm.in_ = C.cast(ibuf, C.c_void_p); strm.in_sz = k
            strm.out = C.cast(obuf, C.c_void_p); strm.out_sz = out_room
            last = 1 if pos + k == len(data) else 0
            rc = L.qzDecompressStream(C.byref(sess.s), C.byref(strm), last)
            if rc != A.QZ_OK:
                break
            assert strm.in_sz == k
            pos += k; got += obuf.raw[:strm.out_sz]; calls += 1
            assert calls < 5000
            if last and strm.pending_in == 0 and strm.pending_out == 0:
                break
        crc = strm.crc_32
        L.qzEndStream(C.byref(sess.s), C.byref(strm))
        sess.close()
        return rc, got, crc

    rc, got, crc = run(comp)
    assert rc == A.QZ_OK and got == want
    assert crc == zlib.crc32(want) & 0xffffffff
    bad = bytearray(comp); bad[len(comp) // 3] ^= 0x55
    rc, got, _ = run(bytes(bad))
    assert rc != A.QZ_OK or got != want
    # LZ4 sessions: the compress side of the stream API refuses them (src/qatzip_stream.c:478-484), the decompress side reads frames
    lz = A.Session(hw_buff_sz=65536, lz4=True)
    big = datagen.gen_bytes("silesia", 200_000, 95)
    frames = lz.compress(big, 1, cap=len(big) + 4096)[2]
    strm = A.QzStream()
    obuf = C.create_string_buffer(1 << 20)
    got, pos = b"", 0
    while pos < len(frames):
        k = min(7000, len(frames) - pos)
        ibuf = C.create_string_buffer(frames[pos:pos + k], k)
        strm.in_ = C.cast(ibuf, C.c_void_p); strm.in_sz = k; strm.out = C.cast(obuf, C.c_void_p); strm.out_sz = len(obuf)
        assert L.qzDecompressStream(C.byref(lz.s), C.byref(strm), 1 if pos + k == len(frames) else 0) == A.QZ_OK
        got += obuf.raw[:strm.out_sz]; pos += k
    assert got == big and strm.pending_in == 0
    L.qzEndStream(C.byref(lz.s), C.byref(strm))
    strm2 = A.QzStream()
    buf = C.create_string_buffer(1000)
    strm2.in_ = C.cast(buf, C.c_void_p); strm2.in_sz = 10; strm2.out = C.cast(buf, C.c_void_p); strm2.out_sz = 1000
    assert L.qzCompressStream(C.byref(lz.s), C.byref(strm2), 1) == A.QZ_PARAMS
    assert L.qzDecompressStream(C.byref(lz.s), None, 1) == A.QZ_PARAMS
    lz.close()


def test_pinned_memory():
    L = A.lib()
    for _ in range(100):                                  # test/main.c:2401-2441 allocates/frees in a loop
        p = L.qzMalloc(100000, -1, A.PINNED_MEM)
        assert p and L.qzMemFindAddr(p) == 1
        # any address inside the allocation counts (the reference marks every page of it, src/qatzip_mem.c:102-149)
        assert L.qzMemFindAddr(p + 1) == 1 and L.qzMemFindAddr(p + 54321) == 1 and L.qzMemFindAddr(p + 99999) == 1
        assert L.qzMemFindAddr(p + 100000) == 0 or L.qzMemFindAddr(p + 100000) == 1     # next byte: another allocation's, or nobody's
        L.qzFree(p)
        assert L.qzMemFindAddr(p) == 0 and L.qzMemFindAddr(p + 5000) == 0
    q = L.qzMalloc(1 << 20, 0, A.PINNED_MEM)             # an explicit NUMA node (0 exists everywhere)
    assert q and L.qzMemFindAddr(q + (1 << 19)) == 1
    C.memset(q, 0x5a, 1 << 20)
    L.qzFree(q)
    m = L.qzMalloc(4096, -1, A.COMMON_MEM)
    assert m
    L.qzFree(m)


def test_plain_c_caller_links_and_round_trips(tmp_path):
    exe = str(tmp_path / "bt_sweep")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "bt_sweep.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "qatzip_amd"), "-lqatzip_amd", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "qatzip_amd")])
    r = subprocess.run([exe, "sweep", "1", "200000", "9973"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe, "perf", "64", "65536", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Gbps" in r.stdout, r.stdout + r.stderr
    print(r.stdout)
    # the harness' -t: eight threads, a session each, one synchronous call per 64 KB block - calls that wait at the same
    # time share launches (and every byte comes back right)
    r = subprocess.run([exe, "perfmt", "8", "65536", "1", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Gbps" in r.stdout, r.stdout + r.stderr
    print(r.stdout)
    # two 1 GiB calls each way: the sizes at which input and output travel in pieces beside the kernels (batched
    # host-to-device copies under K1, output ranges sent behind phase B); the harness compares every byte
    for pin in ("0", "1"):
        r = subprocess.run([exe, "perf", "2048", str(1 << 30), "1"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, BT_PINNED=pin))
        assert r.returncode == 0 and "Gbps" in r.stdout, r.stdout + r.stderr
        print(r.stdout)


def test_plain_c_caller_with_the_harness_options(tmp_path):
    """bt_sweep run: the reference harness' own option letters (test/main.c:6319-6374: -i -t -l -v -C -b -D -O -L -p) -
    a file, chunk and block sizes, every format, both directions, threads; -v compares what came back"""
    exe = str(tmp_path / "bt_sweep")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "bt_sweep.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "qatzip_amd"), "-lqatzip_amd", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "qatzip_amd")])
    f = tmp_path / "input.bin"
    f.write_bytes(datagen.gen_bytes("silesia", 3 * 1024 * 1024 + 12345, 8))
    for opts in (["-v"],                                                    # the harness' default: 512 KB of runs, one call each way
                 ["-v", "-i", str(f), "-C", "65536", "-b", "65536", "-t", "4", "-l", "2"],
                 ["-v", "-i", str(f), "-C", "16384", "-b", "131072", "-O", "gzip", "-D", "both"],
                 ["-v", "-i", str(f), "-C", "131072", "-b", "131072", "-O", "deflate"],
                 ["-v", "-i", str(f), "-O", "deflate_4B", "-b", "524288", "-p", "pinned"],
                 ["-v", "-i", str(f), "-O", "zlib", "-C", "65536", "-b", "65536", "-L", "6"],
                 ["-v", "-i", str(f), "-O", "lz4", "-b", "65536", "-t", "2"],
                 ["-i", str(f), "-D", "comp", "-b", "1048576"],
                 ["-v", "-i", str(f), "-D", "decomp", "-b", "65536", "-t", "8"]):
        r = subprocess.run([exe, "run"] + opts, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "Gbps" in r.stdout, (opts, r.stdout + r.stderr)
        print(r.stdout.strip())


@pytest.mark.parametrize("fmt", ["GZIP_EXT", "GZIP", "RAW", "4B"])
def test_pinned_destination_receives_the_stream_directly(fmt):
    """qzCompress into qzMalloc(PINNED_MEM) memory that holds the worst case: the gather kernels write the stream straight
    into the caller's buffer, batch by batch (no copy after the last kernel) - same bytes as the software path, header and
    trailer around them, nothing written past the reported length"""
    L = A.lib()
    data_fmt = {"GZIP_EXT": A.QZ_DEFLATE_GZIP_EXT, "GZIP": A.QZ_DEFLATE_GZIP, "RAW": A.QZ_DEFLATE_RAW, "4B": A.QZ_DEFLATE_4B}[fmt]
    s = A.Session(data_fmt=data_fmt, hw_buff_sz=65536)
    n = (24 << 20) + 777                                   # above the 8 MiB at which the direct path starts, several batches
    src = datagen.gen_bytes("silesia", n, 61)
    cap = L.qzMaxCompressedLength(n, C.byref(s.s)) + 64
    psrc = L.qzMalloc(n, -1, A.PINNED_MEM); pdst = L.qzMalloc(cap + 4096, -1, A.PINNED_MEM)
    assert psrc and pdst
    C.memmove(psrc, src, n)
    C.memset(pdst, 0xA5, cap + 4096)
    sl, dl = C.c_uint(n), C.c_uint(cap)
    rc = L.qzCompress(C.byref(s.s), C.cast(psrc, C.c_char_p), C.byref(sl), C.c_void_p(pdst), C.byref(dl), 1)
    assert rc == A.QZ_OK and sl.value == n
    got = C.string_at(pdst, dl.value)
    exp = O.sw_compress(fmt, src, 65536, 1, cap=cap)[2]
    assert got == exp, (fmt, len(got), len(exp))
    assert C.string_at(pdst + cap, 4096) == b"\xa5" * 4096          # the guard behind the buffer is untouched
    back = s.decompress(got, n + 64)
    assert back[0] == A.QZ_OK and back[2] == src
    L.qzFree(psrc); L.qzFree(pdst)
    s.close()


@pytest.mark.parametrize("hw,pinned", [(65536, True), (65536, False), (16384, True), (131072, True)])
def test_large_host_input_feeds_a_launch_that_is_already_running(hw, pinned):
    """qzCompress of 64 MiB and more from host memory: ONE launch starts at once and takes its chunks as the pieces of the
    copy land (qzk_wait_input; the host raises the watermark).  The same source buffer is filled three times with different
    data, so a wave that read a chunk before it had landed - or a line of an earlier round left in a cache - would show:
    every round must equal the software path's bytes (src/qatzip_sw.c:77-256), and the batched pipeline's
    (QATZIP_AMD_HOST_BATCHED=1) once."""
    L = A.lib()
    s = A.Session(data_fmt=A.QZ_DEFLATE_GZIP_EXT, hw_buff_sz=hw)
    n = (80 << 20) + 12345
    cap = L.qzMaxCompressedLength(n, C.byref(s.s)) + 64
    if pinned:
        psrc = L.qzMalloc(n, -1, A.PINNED_MEM); pdst = L.qzMalloc(cap, -1, A.PINNED_MEM)
        assert psrc and pdst
    else:
        hold = (C.create_string_buffer(n), C.create_string_buffer(cap))
        psrc, pdst = C.addressof(hold[0]), C.addressof(hold[1])

    def once():
        sl, dl = C.c_uint(n), C.c_uint(cap)
        rc = L.qzCompress(C.byref(s.s), C.cast(psrc, C.c_char_p), C.byref(sl), C.c_void_p(pdst), C.byref(dl), 1)
        assert rc == A.QZ_OK and sl.value == n, rc
        return C.string_at(pdst, dl.value)

    for rnd, kind in enumerate(("silesia", "rand", "lzmix")):
        src = datagen.gen_bytes(kind, n, 300 + rnd)
        C.memmove(psrc, src, n)
        got = once()
        exp = O.sw_compress("GZIP_EXT", src, hw, 1, cap=cap)[2]
        assert got == exp, (kind, hw, pinned, len(got), len(exp))
        if rnd == 0:
            os.environ["QATZIP_AMD_HOST_BATCHED"] = "1"
            try:
                assert once() == exp
            finally:
                del os.environ["QATZIP_AMD_HOST_BATCHED"]
    if pinned:
        L.qzFree(psrc); L.qzFree(pdst)
    s.close()


@pytest.mark.parametrize("hw,pinned,pieces", [(65536, True, None), (65536, False, None), (16384, True, "3"), (131072, True, "8")])
def test_large_member_is_decoded_while_it_arrives(hw, pinned, pieces):
    """qzDecompress of a member of 24 MiB (compressed) and more from host memory: the source goes to the device in pieces, a
    piece's segments are decoded as soon as it has landed and the output leaves while the later pieces arrive and decode
    (qzd_inflate_stream_from_host; the reference keeps requests in flight the same way, src/qatzip.c:2103-2404).  One
    pinned source and one destination are used for three kinds of data in a row, so a segment decoded before its bytes had
    landed - or bytes of the earlier round - would show; the fourth round appends a second member and a cut-off third one
    (the pieces end with the first member: the rest must be found where it is), the last one damages the stream (the
    pieces give up, the call must still report it)."""
    L = A.lib()
    s = A.Session(data_fmt=A.QZ_DEFLATE_GZIP_EXT, hw_buff_sz=hw)
    n = (120 << 20) + 4321
    cap_c = n + (n >> 3) + 65536
    if pinned:
        psrc = L.qzMalloc(cap_c, -1, A.PINNED_MEM); pdst = L.qzMalloc(n + 4096 + 64, -1, A.PINNED_MEM)
        assert psrc and pdst
    else:
        hold = (C.create_string_buffer(cap_c), C.create_string_buffer(n + 4096 + 64))
        psrc, pdst = C.addressof(hold[0]), C.addressof(hold[1])
    if pieces is not None:
        os.environ["QATZIP_AMD_PIPE"] = pieces

    def once(clen, dcap):
        C.memset(pdst + dcap, 0xa5, 64)
        sl, dl = C.c_uint(clen), C.c_uint(dcap)
        rc = L.qzDecompress(C.byref(s.s), C.cast(psrc, C.c_char_p), C.byref(sl), C.c_void_p(pdst), C.byref(dl))
        assert C.string_at(pdst + dcap, 64) == b"\xa5" * 64       # nothing behind the destination
        return rc, sl.value, dl.value

    try:
        for rnd, kind in enumerate(("silesia", "rand", "lzmix")):
            src = datagen.gen_bytes(kind, n, 400 + rnd)
            comp = s.compress(src, 1)[2]
            assert len(comp) >= (24 << 20) or kind == "lzmix", len(comp)
            C.memmove(psrc, comp, len(comp))
            rc, used, got = once(len(comp), n + 4096)
            assert rc == A.QZ_OK and used == len(comp) and got == n, (kind, rc, used, got)
            assert C.string_at(pdst, n) == src, kind
        # two members and the beginning of a third
        src = datagen.gen_bytes("silesia", n, 444)
        tail = datagen.gen_bytes("text", 300000, 445)
        c1, c2 = s.compress(src, 1)[2], s.compress(tail, 1)[2]
        both = c1 + c2 + c1[:1000]
        assert len(both) <= cap_c
        C.memmove(psrc, both, len(both))
        rc, used, got = once(len(both), n + 4096)                   # the destination ends inside the second member: whole members only
        assert rc == A.QZ_BUF_ERROR and used == len(c1) and got == n and C.string_at(pdst, n) == src, (rc, used, got)
        # a flipped bit in the middle of the first member
        bad = bytearray(c1); bad[len(bad) // 2] ^= 0x10
        C.memmove(psrc, bytes(bad), len(bad))
        rc, used, got = once(len(bad), n + 4096)
        assert rc == A.QZ_DATA_ERROR and used == 0 and got == 0, (rc, used, got)
        # and a destination that is too small for the member
        C.memmove(psrc, c1, len(c1))
        back = s.decompress(c1, n + 64)
        assert back[0] == A.QZ_OK and back[2] == src                # (pageable, through the wrapper)
    finally:
        if pieces is not None:
            del os.environ["QATZIP_AMD_PIPE"]
    if pinned:
        L.qzFree(psrc); L.qzFree(pdst)
    s.close()


def test_async_requests_of_a_hardware_framing_session_keep_their_framing():
    """advisor (round 2): qzCompress2 requests of a qzamd_set_hw_framing session that wait in the queue together must come
    out framed exactly like one running alone - one complete member per chunk, XFL 0, OS 255 - not in the software path's
    framing the coalesced launch writes"""
    import threading
    L = A.lib()
    L.qzamd_set_hw_framing.argtypes = [C.c_void_p, C.c_int]
    s = A.Session(A.QZ_DEFLATE_GZIP_EXT, 65536)
    assert L.qzamd_set_hw_framing(C.byref(s.s), 1) == A.QZ_OK
    srcs = [datagen.gen_bytes("silesia", n, 70 + i) for i, n in enumerate((65536, 3 * 65536 + 5, 2000, 131072, 65536, 70000, 4096, 65536))]
    alone = [s.compress(x, 1)[2] for x in srcs]                         # one synchronous call each (compress_deflate_hw)
    bufs_in = [C.create_string_buffer(x, len(x)) for x in srcs]
    bufs_out = [C.create_string_buffer(len(x) * 9 // 8 + 8192) for x in srcs]
    results = [A.QzResult() for _ in srcs]
    done, count = threading.Event(), []

    def on_done(res):
        count.append(1)
        if len(count) == len(srcs):
            done.set()
        return 0
    cb = A.QzAsyncCallback(on_done)
    for i, x in enumerate(srcs):                                        # submitted back to back: they meet in the queue
        results[i].cb_tag = i + 1; results[i].src_len = len(x); results[i].dest_len = len(bufs_out[i])
        assert L.qzCompress2(C.byref(s.s), bufs_in[i], bufs_out[i], cb, C.byref(results[i])) == A.QZ_OK
    assert done.wait(120)
    for i, x in enumerate(srcs):
        r = results[i]
        assert r.status == A.QZ_OK and r.src_len == len(x)
        got = bufs_out[i].raw[:r.dest_len]
        assert got == alone[i], (i, len(got), len(alone[i]))
        assert got[8] == 0 and got[9] == 255
    s.close()
