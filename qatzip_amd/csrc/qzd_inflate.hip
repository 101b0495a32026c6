/*
 * qzd_inflate.hip — host side of the decompress half of the device C ABI.
 *
 * The software path's inflate loop (qzSWDecompressMulti / qzDeflateSWDecompress,
 * src/qatzip_sw.c:258-441,659-695) walks ONE z_stream per member serially.  On the
 * MI355X the same member is cut into independent segments at the byte-aligned
 * Z_FULL_FLUSH markers the compress side leaves after every hw_buff_sz chunk
 * (src/qatzip_sw.c:182-186), every segment is inflated by its own wave (K3), and
 * the host validates that the segments form one contiguous chain that ends in
 * BFINAL.  Marker candidates that are not real boundaries (00 00 FF FF can occur
 * inside compressed data) are discarded by that chain walk.
 *
 * Strategy per stream (qzd_inflate_stream):
 *   1. marker scan (GPU) -> sorted candidate starts; every candidate knows where the next one begins, i.e. how long
 *      it is at most compressed
 *   2a. phase A (Huffman decoding into token streams, K lanes per candidate: qzk_inflate_spec.h) over EVERY candidate,
 *      the chain walk on the host (a candidate that is no boundary drops out, every real segment gets its output
 *      offset), phase B (qzk_lz_resolve_kernel) for the real ones, range by range with the output leaving for the host
 *      behind it when the caller wants it there
 *   2b. (QATZIP_AMD_INFLATE=wave) optimistic single pass, a wave per candidate: candidate k is assumed real and to
 *      produce exactly seg_hint bytes (the session's hw_buff_sz) at k*seg_hint; validated afterwards
 *   3. otherwise two passes: count-only decode of all candidates -> chain walk + prefix sums on the host -> decode of
 *      the true segments at exact offsets
 *   4. streams whose segments reference earlier history (Z_SYNC_FLUSH producers) or have no markers are decoded by
 *      one wave straight through.
 * A member that is still in host memory (qzd_inflate_stream_from_host) goes through 1 and 2a piece by piece while it
 * arrives; whatever the pieces cannot take falls back to the list above.
 */
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string.h>
#include <string>
#include <vector>

#include "qzd_internal.h"
#include <stdlib.h>
#include "qzk_inflate.h"
#include "qzk_inflate_lane.h"
#include "qzk_inflate_spec.h"
#include "qzk_checksum.h"

/* Which kernel for how many segments (round 4, profiles/r4_inflate_crossover.txt; segments of the bench data).
 * With a compressed-length hint per segment the two phases with K lanes per segment (qzk_inflate_spec.h) win at every size
 * measured, from one megabyte (16 segments of 64 KB: 4.7 against 9.3 ms) to four gigabytes: a segment's serial chain is cut
 * in K, and one wave per segment (K3) takes ~9-12 ms for a 64 KB segment however few there are.  Without hints (a foreign
 * stream's flush points, count-only passes) phase A is one lane per segment, and that wins from ~1000 segments on (K3: 17 ms
 * at 4096 segments of 64 KB, 27 at 8192 - the CUs' scalar units; 128 KB segments: 24 ms either way at 2048, 30 against 55 at 8192). */
#define QZD_LANE_MIN_SEGS 1024u
#define QZD_LANE_MIN_SEGS_BIG 2048u
#define QZD_LANE_MIN(seg_bytes) ((seg_bytes) <= 65536u + 64u ? QZD_LANE_MIN_SEGS : QZD_LANE_MIN_SEGS_BIG)
#define QZD_LANE_SEGS_PER_WAVE 16u
/* lanes per segment of phase A (QATZIP_AMD_INFLATE_K = 1, 4, 8, 16, 32 overrides; 1 = the serial phase A).  The lanes
 * of a segment share its tables in LDS, start at evenly spaced bits and fall into step with each other: more lanes, shorter
 * chains and more waves, but more bits decoded twice.  Round 6 measured the table again with three waves a SIMD to choose from
 * (profiles/r6_phaseA_order.txt, phase A in ms at 2 / 3 waves a SIMD):
 *     segments of 64 KB      four lanes     eight          sixteen        thirty-two
 *      1024 ( 64 MiB)         8.9            5.7            3.8 / 3.7      - / 3.4
 *      8192 (512 MiB)        12.2            7.8            6.0 / 5.8      - / 6.5
 *     16384 (  1 GiB)        12.0            8.0            6.2 / 6.4      - / 8.9
 *     32768 (  2 GiB)        13.0           10.3 / 10.3    10.0 / 8.7      - / 12.8
 *     65536 (  4 GiB)        17.6           18.7 / 15.6    21.1 / 17.3     - / 22.3
 *   1 GiB of 128 KB: sixteen 10.5; of 512 KB: thirty-two 18.9 (sixteen 35); of 16 KB (65536 segments): four 5.8, eight 7.4.
 * So: segments above 16 KB take sixteen lanes up to 32768 of them (thirty-two when they are above 128 KB and few) and eight
 * beyond; segments of 16 KB and less - sixteen lanes would get under a hundred bytes each - eight up to 16384, four beyond.
 * (Round 4 kept sixteen lanes from segments above 256 KB: twenty blocks of sixteen pieces outgrew a piece list of 160 and came
 * back through the serial kernel; the list holds 640 now.) */
#define QZD_SPEC_LANES 4u
#define QZD_SPEC_LANES_FEW 8u
#define QZD_SPEC_FEW_SEGS 16384u
#define QZD_SPEC_FEW_SEGS_BIG 32768u
#define QZD_SPEC_LANES_FEWER 16u
#define QZD_SPEC_FEWER_SEGS 8192u
static uint32_t spec_lanes(const qzk_infseg *hs, uint32_t nsegs)
{
    const bool big = hs[0].out_cap > 16384u + 64u, upto512 = hs[0].out_cap <= 524288u + 64u;
    uint32_t K = nsegs <= QZD_SPEC_FEWER_SEGS && hs[0].out_cap > 131072u + 64u && upto512 ? 32u
               : big && upto512 && nsegs <= QZD_SPEC_FEW_SEGS_BIG ? QZD_SPEC_LANES_FEWER
               : big ? QZD_SPEC_LANES_FEW
               : nsegs <= QZD_SPEC_FEW_SEGS ? QZD_SPEC_LANES_FEW : QZD_SPEC_LANES;
    const char *ke = getenv("QATZIP_AMD_INFLATE_K");
    if (ke) { int v = atoi(ke); if (v == 1 || v == 4 || v == 8 || v == 16 || v == 32) K = (uint32_t)v; }
    /* it needs a compressed-length hint (qzk_infseg.pad) and segments that write output and begin with no history */
    for (uint32_t i = 0; i < nsegs && K > 1; i++)
        if ((hs[i].flags & (QZK_INF_COUNT_ONLY | QZK_INF_THROUGH_FLUSH)) || hs[i].pad == 0) K = 1;
    return K;
}
/* two phases or a wave per segment? (QATZIP_AMD_INFLATE = lane / wave overrides) */
static bool use_lanes(uint32_t K, uint32_t nsegs, uint32_t seg_bytes)
{
    const char *force = getenv("QATZIP_AMD_INFLATE");
    if (force) return force[0] == 'l';
    return K > 1 || nsegs >= QZD_LANE_MIN(seg_bytes);
}
#define QZD_SO_PARTS 8u             /* output ranges a streamed decode is resolved and sent in */

/* positions p (relative to d_src) such that src[p-4..p) == 00 00 FF FF.  A thread takes sixteen byte positions a trip:
 * five dwords (the last one for the three bytes that reach into the next piece), sixteen funnel shifts - a quarter of
 * the load instructions of one dword per position (0.74 -> ~0.3 ms for the 0.8 GB of a 2 GiB call). */
#ifdef QZK_SPEC_PROF
extern __device__ unsigned long long qzk_stamp[8];
#endif
__global__ void qzk_marker_kernel(const uint8_t *src, uint64_t n, uint32_t *list, uint32_t cap, uint32_t *count)
{
#ifdef QZK_SPEC_PROF
    if ((threadIdx.x & 63) == 0) { atomicMin(&qzk_stamp[0], (unsigned long long)__builtin_amdgcn_s_memrealtime()); }
#endif
    /* a lane takes the sixteen positions [b, b + 16): their dwords are the sixteen bytes it loads (two 8-byte loads, any
     * alignment) and the first dword of the NEXT lane's sixteen, which comes through a cross-lane read - the wave's last lane
     * loads it.  Bytes behind n read as 0.  (Round 5; five dword loads a lane before.  The kernel's time did not move -
     * 0.79 ms for the 1.6 GB of a 4 GiB call, 2 TB/s, either way: a wave has one trip's loads in flight and waits for them,
     * it is the latency of a trip that bounds it, not the number of load instructions.  More trips in flight is what is left.) */
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t b0 = ((uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * 16; b0 + 4 <= n; b0 += stride) {      /* (the same trips for a wave's lanes) */
        const uint64_t b = b0 + (uint64_t)lane * 16;
        uint32_t d[5];
        if (b + 16 <= n) {
            const uint64_t lo = qzk_ld64u(src + b), hi = qzk_ld64u(src + b + 8);
            d[0] = (uint32_t)lo; d[1] = (uint32_t)(lo >> 32); d[2] = (uint32_t)hi; d[3] = (uint32_t)(hi >> 32);
        } else {
            for (int k = 0; k < 4; k++) {
                d[k] = 0;
                for (int j = 0; j < 4; j++) if (b + 4 * k + j < n) d[k] |= (uint32_t)src[b + 4 * k + j] << (8 * j);
            }
        }
        d[4] = qz_shfl(d[0], (int)((lane + 1u) & 63u));
        if (lane == 63u) {
            d[4] = 0;
            if (b + 20 <= n) d[4] = qz_ld32(src + b + 16);
            else for (int j = 0; j < 4; j++) if (b + 16 + j < n) d[4] |= (uint32_t)src[b + 16 + j] << (8 * j);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t w = (k & 3) ? __builtin_amdgcn_alignbyte(d[k / 4 + 1], d[k / 4], (uint32_t)(k & 3)) : d[k / 4];
            if (w == 0xFFFF0000u && b + k + 4 <= n) {
                uint32_t i = atomicAdd(count, 1u);
                if (i < cap) list[i] = (uint32_t)(b + k + 4);
            }
        }
    }
}
/* the scan as it was through round 4 (five dword loads a lane): QATZIP_AMD_MARKER_CHECK=1 runs it beside the one above and
 * compares what they find (tests/test_gpu_inflate.py) */
__global__ void qzk_marker_ref_kernel(const uint8_t *src, uint64_t n, uint32_t *list, uint32_t cap, uint32_t *count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16;
    for (uint64_t b = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; b + 4 <= n; b += stride) {
        uint32_t d[5];
        for (int k = 0; k < 5; k++) {
            d[k] = 0;
            if (b + 4 * k + 4 <= n) d[k] = qz_ld32(src + b + 4 * k);
            else for (int j = 0; j < 4; j++) if (b + 4 * k + j < n) d[k] |= (uint32_t)src[b + 4 * k + j] << (8 * j);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t w = (k & 3) ? __builtin_amdgcn_alignbyte(d[k / 4 + 1], d[k / 4], (uint32_t)(k & 3)) : d[k / 4];
            if (w == 0xFFFF0000u && b + k + 4 <= n) {
                uint32_t i = atomicAdd(count, 1u);
                if (i < cap) list[i] = (uint32_t)(b + k + 4);
            }
        }
    }
}

/* Control data between the host's pinned staging and the device (segment records in, results and candidate lists out):
 * moved by a small KERNEL, not by the copy engines.  A copy engine takes its copies in order, so a 16-byte result queued
 * behind a quarter of a gigabyte of output on its way to the host arrived when that had arrived - the piece-wise decode
 * (qzd_inflate_stream_from_host) stood still behind its own output for 30 ms.  Both ends are multiples of four bytes;
 * the host side is hipHostMalloc memory, which the device reads and writes in place. */
__global__ void qzk_ctl_copy_kernel(uint32_t *dst, const uint32_t *src, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
/* The sub-stream descriptors of a call whose segments are all of one size (every call of fixed hw_buff_sz chunks): segment i's
 * lane j has the region [i * seg + pre[j], + len[j]) - written here instead of being built on the host and sent across (sixteen
 * bytes a lane: 8 MB for the bench's 4 GiB call, a third of a millisecond of PCIe and as much of host loop before phase A). */
struct qzk_tsfill { uint64_t pre[32], len[32]; uint64_t seg; };
__global__ void qzk_ts_fill_kernel(qzk_tokseg *ts, uint32_t nsegs, uint32_t K, const qzk_tsfill F)
{
    const size_t n = (size_t)nsegs * K;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const uint64_t i = t / K, j = t % K, a = i * F.seg + F.pre[j];
        ts[t].lit_off = a; ts[t].seq_off = (a + F.len[j]) / 8;
    }
}
static hipError_t ctl_copy(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    const size_t n4 = (bytes + 3) / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 256);
    hipLaunchKernelGGL(qzk_ctl_copy_kernel, dim3(blocks), dim3(256), 0, st, (uint32_t *)dst, (const uint32_t *)src, n4);
    return hipGetLastError();
}

/* Two-phase inflate of nsegs segments (qzk_inflate_lane.h / qzk_inflate_spec.h).  K == 1: one lane decodes a whole
 * segment; K > 1: K lanes per segment decode speculatively, and whatever that kernel hands back (QZK_INF_ESPEC: stored
 * or several blocks, bad data, ...) goes through the K == 1 launch afterwards.  h_res receives every result. */
static int two_phase(qzd_ctx *c, const uint8_t *d_comp, uint8_t *d_out, const qzk_infseg *hs, uint32_t nsegs,
                     qzk_infres *h_res, uint32_t K, hipStream_t st, bool run_b = true)
{
    const size_t sb = (size_t)nsegs * sizeof(qzk_infseg), rb = (size_t)nsegs * sizeof(qzk_infres);
    /* everything that crosses PCIe here goes through the pinned mirror of the aux scratch (a copy from or to pageable
     * memory is staged by the runtime in pieces, each with its own wait: a quarter of a millisecond per megabyte, four
     * times per call, between two kernels) */
    const size_t o_res = (sb + 15) & ~(size_t)15, o_ts = o_res + ((rb + 15) & ~(size_t)15);
    const size_t o_ord = o_ts + (((size_t)nsegs * K * sizeof(qzk_tokseg) + 15) & ~(size_t)15);
    int rc = qzd_aux_reserve(c, o_ord + (size_t)nsegs * 4 + 64);
    if (rc) return rc;
    qzk_infseg *d_segs = (qzk_infseg *)c->d_aux;
    qzk_infres *d_res = (qzk_infres *)(c->d_aux + o_res);
    qzk_infseg *st_segs = (qzk_infseg *)c->h_aux;
    qzk_infres *st_res = (qzk_infres *)(c->h_aux + o_res);
    qzk_tokseg *tsv = (qzk_tokseg *)(c->h_aux + o_ts);
    uint32_t *st_ord = (uint32_t *)(c->h_aux + o_ord);
    memcpy(st_segs, hs, sb);
    /* every sub-stream is one region of the scratch, literals up from its first byte, sequences down from its end
     * (qzk_inflate_lane.h).  K == 1: a region that holds any segment; K > 1: regions sized by need (qzk_inflate_spec.h), and
     * behind them R whole-segment regions for the segments that kernel hands back to the serial one */
    uint64_t arena = 0, max_cap = 0;
    /* one size for all (no count-only segments, K <= 32): the descriptors are written on the device (qzk_ts_fill_kernel), and the host's
     * copy of them only if a segment is handed back */
    bool uniform = nsegs > 0 && K <= 32;
    for (uint32_t i = 0; i < nsegs && uniform; i++) uniform = !(hs[i].flags & QZK_INF_COUNT_ONLY) && hs[i].out_cap == hs[0].out_cap;
    qzk_tsfill F;
    auto fill_host = [&]() {                                    /* the same layout in the host's table */
        for (uint32_t i = 0; i < nsegs; i++)
            for (uint32_t j = 0; j < K; j++) {
                const uint64_t a = (uint64_t)i * F.seg + F.pre[j];
                tsv[(size_t)i * K + j].lit_off = a; tsv[(size_t)i * K + j].seq_off = (a + F.len[j]) / 8;
            }
    };
    if (uniform) {
        F.seg = 0; max_cap = hs[0].out_cap;
        for (uint32_t j = 0; j < K; j++) {
            F.len[j] = K == 1 ? QZK_TOK_REGION(hs[0].out_cap) : QZK_SPEC_REGION(hs[0].out_cap, K, j);
            F.pre[j] = F.seg; F.seg += F.len[j];
        }
        arena = (uint64_t)nsegs * F.seg;
    } else
    for (uint32_t i = 0; i < nsegs; i++) {
        const bool writes = !(hs[i].flags & QZK_INF_COUNT_ONLY);
        if (writes && hs[i].out_cap > max_cap) max_cap = hs[i].out_cap;
        for (uint32_t j = 0; j < K; j++) {
            const uint64_t rg = !writes ? 0 : K == 1 ? QZK_TOK_REGION(hs[i].out_cap) : QZK_SPEC_REGION(hs[i].out_cap, K, j);
            tsv[(size_t)i * K + j].lit_off = arena; tsv[(size_t)i * K + j].seq_off = (arena + rg) / 8;
            arena += rg;
        }
    }
    const uint32_t R = K == 1 ? 0u : std::min<uint32_t>(nsegs, QZK_SPEC_HANDBACK(nsegs));
    const uint64_t hb0 = arena, hb_rg = QZK_TOK_REGION(max_cap);
    arena += (uint64_t)R * hb_rg;
    const size_t tabb = ((size_t)nsegs * sizeof(qzk_inf_tab) + 255) & ~(size_t)255;
    const size_t tsb = ((size_t)nsegs * K * sizeof(qzk_tokseg) + 255) & ~(size_t)255;
    const size_t chb = ((size_t)nsegs * sizeof(qzk_chain) + 255) & ~(size_t)255;
    const size_t rcb = K == 1 ? 0 : (((size_t)nsegs * K * QZK_SPEC_NREC * sizeof(qzk_rec) + 255) & ~(size_t)255);
    const size_t litb = (arena + 1023) & ~(uint64_t)255, seqb = 0;             /* (one arena; phase B reads whole 16-byte rows past a region's literals) */
    /* output streaming: only the plain decode of a whole member (every segment writes, known output offsets) */
    const bool stream_out = run_b && c->so_host && c->so_nat && nsegs >= QZD_LANE_MIN_SEGS;
    const size_t ordb = ((size_t)nsegs * 4 + 255) & ~(size_t)255;
    const size_t need = tabb + tsb + chb + rcb + litb + seqb + ordb + 256;
    /* the scratch grows with the largest call and shrinks again when EIGHT calls in a row have needed less than a quarter of it
     * (a 4 GiB decode leaves gigabytes behind; a session of small calls gives them back, one that alternates large and small
     * ones keeps them - every change is a free and an allocation).  What runs on the scratch runs on this context's
     * streams: they are waited for, not the device - helper contexts decode other pieces meanwhile */
    if (c->big_cap > ((size_t)1 << 30) && need * 4 < c->big_cap) c->big_small++; else c->big_small = 0;
    if (need > c->big_cap || c->big_small >= 8) {
        c->big_small = 0;
        hipStreamSynchronize(st); hipStreamSynchronize(c->st_out);
        if (c->d_big) hipFree(c->d_big);
        c->d_big = NULL; c->big_cap = 0;
        /* QATZIP_AMD_SCRATCH_MAX=<bytes>: a ceiling for this scratch (tests: the paths a failed allocation takes) */
        const char *cap_e = getenv("QATZIP_AMD_SCRATCH_MAX");
        const bool refused = cap_e && need > (size_t)strtoull(cap_e, NULL, 10);
        if (refused || hipMalloc(&c->d_big, need) != hipSuccess) {
            (void)hipGetLastError();
            c->d_big = NULL;
            /* not enough device memory for K sub-streams per segment: one per segment needs a third of it; after that the
             * callers take the wave-per-segment kernel, which needs none */
            if (K > 1) return two_phase(c, d_comp, d_out, hs, nsegs, h_res, 1, st, run_b);
            snprintf(c->err, sizeof(c->err), "inflate: %zu bytes of decode scratch are not available", need);
            return QZD_ERR_NOMEM;
        }
        c->big_cap = need;
    }
    uint8_t *pb = c->d_big;
    qzk_inf_tab *tb_d = (qzk_inf_tab *)pb; pb += tabb;
    qzk_tokseg *ts_d = (qzk_tokseg *)pb; pb += tsb;
    qzk_chain *ch_d = (qzk_chain *)pb; pb += chb;
    qzk_rec *rec_d = (qzk_rec *)pb; pb += rcb;
    uint8_t *lit_d = pb; pb += litb;
    qzk_seq *seq_d = (qzk_seq *)lit_d;                  /* the same arena: a region's sequences count down from its end */
    uint32_t *ord_d = (uint32_t *)pb;
    HIPCHK(c, ctl_copy(d_segs, st_segs, sb, st));
    if (uniform) {
        hipLaunchKernelGGL(qzk_ts_fill_kernel, dim3((unsigned)std::min<size_t>(((size_t)nsegs * K + 255) / 256, 1024)), dim3(256), 0, st, ts_d, nsegs, K, F);
        HIPCHK(c, hipGetLastError());
    } else HIPCHK(c, ctl_copy(ts_d, tsv, (size_t)nsegs * K * sizeof(qzk_tokseg), st));
    HIPCHK(c, hipEventRecord(c->ev[1][1], st));
    /* segments per single-wave workgroup of the serial phase A: each lane keeps 1.25 KiB of root tables in LDS, and partly
     * filled waves give the serial decode loops more waves to hide behind (measured in DESIGN.md K3b) */
    uint32_t lpw = QZD_LANE_SEGS_PER_WAVE;
    const char *le = getenv("QATZIP_AMD_INFLATE_LPW");
    if (le) { int v = atoi(le); if (v == 8 || v == 16 || v == 32 || v == 64) lpw = (uint32_t)v; }
    const char *oe = getenv("QATZIP_AMD_INFLATE_OCC");
    const int occ = oe ? atoi(oe) : 0;
#ifndef QZD_TOK_OCC
#define QZD_TOK_OCC 2           /* waves per SIMD of phase A's sixteen-lane workgroups: what their root tables leave room for in LDS */
#endif
    /* the serial phase A over all segments (order NULL) or over `count` of them picked by index; it fills sub-stream 0 */
    auto tok_launch = [&](const uint32_t *order, uint32_t count) {
        const dim3 grid(((order ? count : nsegs) + lpw - 1) / lpw), blk(lpw);
#define QZD_TOK_LAUNCH(N, W) hipLaunchKernelGGL((qzk_inflate_tok_kernel<N, W>), grid, blk, 0, st, d_comp, d_segs, d_res, nsegs, tb_d, \
                                               ts_d, lit_d, seq_d, ch_d, order, count, K)
        if (lpw == 8) { if (occ == 2) QZD_TOK_LAUNCH(8, 2); else QZD_TOK_LAUNCH(8, 4); }
        else if (lpw == 32) QZD_TOK_LAUNCH(32, 1); else if (lpw == 64) QZD_TOK_LAUNCH(64, 1);
        else QZD_TOK_LAUNCH(16, QZD_TOK_OCC);
#undef QZD_TOK_LAUNCH
    };
    bool res_fresh = false;                                         /* the host's copy of the results is current (nothing ran since it was taken) */
    if (K == 1) tok_launch(NULL, 0);
    else {
        const uint32_t spw = 64 / K;
        const dim3 grid((nsegs + spw - 1) / spw), blk(64);
        static std::atomic<uint64_t> spec_epoch{0};                 /* tags the marks of a launch: scratch left by an earlier one is not mistaken for them */
        const uint64_t epoch = ++spec_epoch;
        /* how far a block's last lane may run beyond its share before the rest is shared out again (a round more): segments
         * above 64 KB hold several blocks of very different length and pay for every lane left alone; small launches end
         * with their longest lane; a full launch of 64 KB segments is better off without the extra rounds
         * (profiles/r4_inflate_crossover.txt) */
        uint32_t over = hs[0].out_cap > 65536u + 64u ? 2u : nsegs <= 4096u ? 1u : 1000u;
        const char *ov = getenv("QATZIP_AMD_INFLATE_OVER");
        if (ov && atoi(ov) > 0) over = (uint32_t)atoi(ov);
        /* waves per SIMD (qzk_inflate_spec.h): three - with a third of the registers spilled - pay from the third round of
         * resident waves on (QATZIP_AMD_INFLATE_OCC=2 / 3 overrides) */
        int socc = grid.x > 16u * c->cus && K != 4 ? 3 : 2;
        if (occ == 2 || (occ == 3 && K != 4)) socc = occ;
#define QZD_SPEC_LAUNCH(N, W) hipLaunchKernelGGL((qzk_inflate_spec_kernel<N, W>), grid, blk, 0, st, d_comp, d_segs, d_res, nsegs, tb_d, \
                                             ts_d, lit_d, seq_d, ch_d, rec_d, epoch, over)
        if (K == 4) QZD_SPEC_LAUNCH(4, 2);
        else if (K == 16) { if (socc == 3) QZD_SPEC_LAUNCH(16, 3); else QZD_SPEC_LAUNCH(16, 2); }
        else if (K == 32) { if (socc == 3) QZD_SPEC_LAUNCH(32, 3); else QZD_SPEC_LAUNCH(32, 2); }
        else { if (socc == 3) QZD_SPEC_LAUNCH(8, 3); else QZD_SPEC_LAUNCH(8, 2); }
#undef QZD_SPEC_LAUNCH
        /* what that kernel hands back (QZK_INF_ESPEC: a sub-stream outgrew its scratch, too many pieces) goes through the
         * serial phase A, into the segment's first sub-stream - which is sized for a whole segment */
        HIPCHK(c, ctl_copy(st_res, d_res, rb, st));
        HIPCHK(c, hipStreamSynchronize(st));
        uint32_t nredo = 0;
        for (uint32_t i = 0; i < nsegs; i++) if (st_res[i].status == QZK_INF_ESPEC) st_ord[nredo++] = i;
        if (nredo && getenv("QATZIP_AMD_TRACE")) {
            uint32_t hist[64] = {0};
            for (uint32_t i = 0; i < nredo; i++) { const uint32_t w = st_res[st_ord[i]].nblocks; hist[w < 64 ? w : 63]++; }
            fprintf(stderr, "[two_phase] K=%u: %u of %u segments handed back to the serial kernel; reasons:", K, nredo, nsegs);
            for (int k = 0; k < 64; k++) if (hist[k]) fprintf(stderr, " %d:%u", k, hist[k]);
            fprintf(stderr, "\n");
        }
        if (nredo > R) return two_phase(c, d_comp, d_out, hs, nsegs, h_res, 1, st, run_b);      /* more than the hand-back area holds: this data is not what K lanes are for */
        res_fresh = nredo == 0;
        if (nredo) {
            if (uniform) fill_host();
            for (uint32_t k = 0; k < nredo; k++) {
                qzk_tokseg &t0 = tsv[(size_t)st_ord[k] * K];
                t0.lit_off = hb0 + (uint64_t)k * hb_rg; t0.seq_off = (hb0 + (uint64_t)(k + 1) * hb_rg) / 8;
            }
            HIPCHK(c, ctl_copy(ts_d, tsv, (size_t)nsegs * K * sizeof(qzk_tokseg), st));
            HIPCHK(c, ctl_copy(ord_d, st_ord, (size_t)nredo * 4, st));
            tok_launch(ord_d, nredo);
        }
    }
    HIPCHK(c, hipEventRecord(c->ev[1][0], st));
    c->tp.segs = d_segs; c->tp.res = d_res; c->tp.ts = ts_d; c->tp.lits = lit_d; c->tp.seqs = seq_d; c->tp.chains = ch_d;
    c->tp.ord = ord_d; c->tp.nsegs = nsegs; c->tp.K = K;
    if (!run_b) {
        /* phase A only: its results say which candidates are real segments and where their output belongs;
         * two_phase_resolve() runs phase B once the host has decided */
        if (!res_fresh) {
            HIPCHK(c, ctl_copy(st_res, d_res, rb, st));
            HIPCHK(c, hipStreamSynchronize(st));
        } else HIPCHK(c, hipEventSynchronize(c->ev[1][0]));         /* (the stream stood still when it was recorded) */
        HIPCHK(c, hipGetLastError());
        memcpy(h_res, st_res, rb);
        float ta = 0;
        if (hipEventElapsedTime(&ta, c->ev[1][1], c->ev[1][0]) == hipSuccess) c->inf_ms[3] += ta;   /* phase A's kernel(s) */
        return QZD_OK;
    }
    if (stream_out) {
        HIPCHK(c, hipStreamSynchronize(st));                        /* the redo list in st_ord may still be on its way to the device */
        memcpy(st_ord, c->so_nat, (size_t)nsegs * 4); HIPCHK(c, ctl_copy(ord_d, st_ord, (size_t)nsegs * 4, st));
    }
    if (!stream_out) {
        hipLaunchKernelGGL(qzk_lz_resolve_kernel, dim3((nsegs + QZK_RES_WAVES - 1) / QZK_RES_WAVES), dim3(64 * QZK_RES_WAVES), 0, st,
                           d_comp, d_out, d_segs, d_res, nsegs, ts_d, K, lit_d, seq_d, ch_d, (const uint32_t *)NULL, 0u);
        HIPCHK(c, hipEventRecord(c->ev[1][2], st));
    } else {
        /* phase B over up to QZD_SO_PARTS ranges of the output, in output order; each range leaves for the host on the copy
         * stream as soon as its launch is done, while the next range is being resolved.  (Whether the ranges are what
         * the caller wanted is decided afterwards, from the results; a decode that turns out wrong is simply copied
         * again as a whole.) */
        const uint32_t parts = std::min<uint32_t>(QZD_SO_PARTS, std::max<uint32_t>(1u, nsegs / 4096u));      /* (as two_phase_resolve) */
        uint64_t off[QZD_SO_PARTS + 1];
        for (uint32_t p = 0; p < parts; p++) {
            const uint32_t first = (uint32_t)((uint64_t)nsegs * p / parts), end = (uint32_t)((uint64_t)nsegs * (p + 1) / parts);
            off[p] = hs[c->so_nat[first]].out_off;
            const qzk_infseg &lastseg = hs[c->so_nat[end - 1]];
            off[p + 1] = lastseg.out_off + lastseg.out_cap;
            hipLaunchKernelGGL(qzk_lz_resolve_kernel, dim3((end - first + QZK_RES_WAVES - 1) / QZK_RES_WAVES), dim3(64 * QZK_RES_WAVES), 0, st,
                               d_comp, d_out, d_segs, d_res, nsegs, ts_d, K, lit_d, seq_d, ch_d, (const uint32_t *)(ord_d + first), end - first);
            HIPCHK(c, hipEventRecord(c->so_ev[p], st));
        }
        HIPCHK(c, hipEventRecord(c->ev[1][2], st));
        for (uint32_t p = 0; p < parts; p++) {       /* a pageable destination makes these block the host: all launches are out already */
            HIPCHK(c, hipStreamWaitEvent(c->st_out, c->so_ev[p], 0));
            HIPCHK(c, hipMemcpyAsync(c->so_host + off[p], d_out + off[p], off[p + 1] - off[p], hipMemcpyDeviceToHost, c->st_out));
        }
        HIPCHK(c, hipStreamSynchronize(c->st_out));
        c->so_sent = off[parts];
    }
    HIPCHK(c, ctl_copy(st_res, d_res, rb, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    memcpy(h_res, st_res, rb);
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev[1][0], c->ev[1][2]) == hipSuccess) c->inf_ms[2] += t;     /* phase B share */
    if (hipEventElapsedTime(&t, c->ev[1][1], c->ev[1][0]) == hipSuccess) c->inf_ms[3] += t;     /* phase A's kernel(s) */
    return QZD_OK;
}

/* Phase B for the segments of the last two_phase(run_b = false) call that the host found to be real: hs = the same
 * records with their final out_off (candidates that are no segments carry QZK_INF_COUNT_ONLY and are skipped),
 * h_order = the real ones in output order.  With h_dst the output leaves for the host range by range behind the launches
 * (off_of / end_of give a range's bytes).  Results (phase B can still find a bad distance) come back in h_res. */
static int two_phase_resolve(qzd_ctx *c, const uint8_t *d_comp, uint8_t *d_out, const qzk_infseg *hs, uint32_t nsegs,
                             const uint32_t *h_order, uint32_t count, qzk_infres *h_res, uint8_t *h_dst, hipStream_t st, hipStream_t out_st = NULL,
                             const std::function<void()> *while_leaving = NULL /* called once the output's copies are queued, before they are waited for */)
{
    if (!out_st) out_st = c->st_out;
    if (nsegs != c->tp.nsegs || count == 0) return QZD_ERR_PARAM;
    const uint32_t K = c->tp.K;                                      /* sub-streams per segment of the phase A that ran */
    qzk_infseg *d_segs = (qzk_infseg *)c->tp.segs; qzk_infres *d_res = (qzk_infres *)c->tp.res;
    /* the pinned mirror, laid out as two_phase() left it (same nsegs, K == 1) */
    const size_t sb = (size_t)nsegs * sizeof(qzk_infseg), rb = (size_t)nsegs * sizeof(qzk_infres);
    const size_t o_res = (sb + 15) & ~(size_t)15, o_ts = o_res + ((rb + 15) & ~(size_t)15);
    const size_t o_ord = o_ts + (((size_t)nsegs * K * sizeof(qzk_tokseg) + 15) & ~(size_t)15);
    if (o_ord + (size_t)nsegs * 4 > c->aux_cap) return QZD_ERR_PARAM;
    qzk_infres *st_res = (qzk_infres *)(c->h_aux + o_res);
    memcpy(c->h_aux, hs, sb);
    memcpy(c->h_aux + o_ord, h_order, (size_t)count * 4);
    HIPCHK(c, ctl_copy(d_segs, c->h_aux, sb, st));
    HIPCHK(c, ctl_copy(c->tp.ord, c->h_aux + o_ord, (size_t)count * 4, st));
    HIPCHK(c, hipEventRecord(c->ev[1][0], st));
    /* ranges of the output that leave one behind the other: a launch lasts as long as its slowest segment (~1 ms for 64 KB),
     * so a range is worth a launch of its own from ~4096 segments on (eight launches over 3000 segments took 8.8 ms, one 1.3) */
    const uint32_t parts = !h_dst ? 1u : std::min<uint32_t>(QZD_SO_PARTS, std::max<uint32_t>(1u, count / 4096u));
    uint64_t off[QZD_SO_PARTS + 1];
    for (uint32_t p = 0; p < parts; p++) {
        const uint32_t first = (uint32_t)((uint64_t)count * p / parts), end = (uint32_t)((uint64_t)count * (p + 1) / parts);
        off[p] = hs[h_order[first]].out_off;
        off[p + 1] = hs[h_order[end - 1]].out_off + hs[h_order[end - 1]].out_cap;
        hipLaunchKernelGGL(qzk_lz_resolve_kernel, dim3((end - first + QZK_RES_WAVES - 1) / QZK_RES_WAVES), dim3(64 * QZK_RES_WAVES), 0, st,
                           d_comp, d_out, d_segs, d_res, nsegs, (const qzk_tokseg *)c->tp.ts, K, (const uint8_t *)c->tp.lits,
                           (const qzk_seq *)c->tp.seqs, (const qzk_chain *)c->tp.chains, (const uint32_t *)(c->tp.ord + first), end - first);
        if (h_dst) HIPCHK(c, hipEventRecord(c->so_ev[p], st));
    }
    HIPCHK(c, hipEventRecord(c->ev[1][2], st));
    if (h_dst) {
        for (uint32_t p = 0; p < parts; p++) {
            HIPCHK(c, hipStreamWaitEvent(out_st, c->so_ev[p], 0));
            HIPCHK(c, hipMemcpyAsync(h_dst + off[p], d_out + off[p], off[p + 1] - off[p], hipMemcpyDeviceToHost, out_st));
        }
        if (while_leaving) (*while_leaving)();
        HIPCHK(c, hipStreamSynchronize(out_st));
        c->so_sent = off[parts];
    }
    HIPCHK(c, ctl_copy(st_res, d_res, rb, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    memcpy(h_res, st_res, rb);
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev[1][0], c->ev[1][2]) == hipSuccess) c->inf_ms[2] += t;
    return QZD_OK;
}

extern "C" int qzd_inflate_segments(qzd_ctx *c, const uint8_t *d_comp, uint8_t *d_out, const void *h_segs,
                                    uint32_t nsegs, void *h_res)
{
    if (!c || !h_segs || !h_res) return QZD_ERR_PARAM;
    if (nsegs == 0) return QZD_OK;
    hipSetDevice(c->device);
    hipStream_t st = c->st[0];
    const qzk_infseg *hs = (const qzk_infseg *)h_segs;
    /* few segments: one wave each; thousands: the two-phase path (the wave-per-segment kernel is bound by the CU's
     * scalar unit, the lane kernels spread the serial work over the vector lanes) */
    const uint32_t K = spec_lanes(hs, nsegs);
    const bool lanes = use_lanes(K, nsegs, hs[0].out_cap);
    HIPCHK(c, hipEventRecord(c->ev[0][0], st));
    bool waves = !lanes;
    if (lanes) {
        int rc = two_phase(c, d_comp, d_out, hs, nsegs, (qzk_infres *)h_res, K, st);
        if (rc == QZD_ERR_NOMEM) waves = true;                      /* no room for the token streams: a wave per segment needs none */
        else if (rc) return rc;
    }
    if (waves) {
        const size_t sb = (size_t)nsegs * sizeof(qzk_infseg), rb = (size_t)nsegs * sizeof(qzk_infres);
        int rc = qzd_aux_reserve(c, sb + rb + 64);
        if (rc) return rc;
        qzk_infseg *d_segs = (qzk_infseg *)c->d_aux;
        qzk_infres *d_res = (qzk_infres *)(c->d_aux + ((sb + 15) & ~(size_t)15));
        HIPCHK(c, hipMemcpyAsync(d_segs, h_segs, sb, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(qzk_inflate_kernel, dim3((nsegs + QZK_INF_WAVES - 1) / QZK_INF_WAVES), dim3(64 * QZK_INF_WAVES),
                           0, st, d_comp, d_out, d_segs, d_res, nsegs);
        HIPCHK(c, hipMemcpyAsync(h_res, d_res, rb, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev[0][1], st));
    HIPCHK(c, hipStreamSynchronize(st));
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev[0][0], c->ev[0][1]) == hipSuccess) c->inf_ms[0] += t;
    return QZD_OK;
}

extern "C" int qzd_crc32_ranges(qzd_ctx *c, const uint8_t *d_data, const void *h_ranges, uint32_t nranges,
                                uint32_t *h_crc)
{
    if (!c || !h_ranges || !h_crc) return QZD_ERR_PARAM;
    if (nranges == 0) return QZD_OK;
    hipSetDevice(c->device);
    const size_t rb = (size_t)nranges * sizeof(qzk_range), cb = (size_t)nranges * 4;
    int rc = qzd_aux_reserve(c, rb + cb + 64);
    if (rc) return rc;
    qzk_range *d_r = (qzk_range *)c->d_aux;
    const size_t o_c = (rb + 15) & ~(size_t)15;
    uint32_t *d_c = (uint32_t *)(c->d_aux + o_c);
    hipStream_t st = c->st[0];
    memcpy(c->h_aux, h_ranges, rb);                                 /* through the pinned mirror, both ways */
    HIPCHK(c, hipMemcpyAsync(d_r, c->h_aux, rb, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev[0][2], st));
    hipLaunchKernelGGL(qzk_crc_kernel, dim3(nranges), dim3(QZK_HT), 0, st, d_data, d_r, nranges, d_c);
    HIPCHK(c, hipEventRecord(c->ev[0][3], st));
    HIPCHK(c, hipMemcpyAsync(c->h_aux + o_c, d_c, cb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    memcpy(h_crc, c->h_aux + o_c, cb);
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev[0][2], c->ev[0][3]) == hipSuccess) c->inf_ms[1] += t;
    return QZD_OK;
}

/* zlib adler32_combine() */
extern "C" uint32_t qzd_adler32_combine(uint32_t ad1, uint32_t ad2, uint64_t len2)
{
    const uint64_t M = 65521;
    const uint64_t a1 = ad1 & 0xffff, b1 = ad1 >> 16, a2 = ad2 & 0xffff, b2 = ad2 >> 16;
    const uint64_t a = (a1 + a2 + M - 1) % M;
    const uint64_t b = (b1 + b2 + (len2 % M) * ((a1 + M - 1) % M)) % M;
    return (uint32_t)(b << 16 | a);
}

/* Adler-32 of every chunk_sz chunk of d_data[0..n) (h_adler: one value per chunk; fold with qzd_adler32_combine) */
extern "C" int qzd_adler32_chunks(qzd_ctx *c, const uint8_t *d_data, uint64_t n, uint32_t chunk_sz, uint32_t *h_adler)
{
    if (!c || !h_adler || (n && !d_data) || chunk_sz == 0 || chunk_sz > 512 * 1024) return QZD_ERR_PARAM;
    const uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    hipSetDevice(c->device);
    int rc = qzd_aux_reserve(c, (size_t)nchunks * 4 + 64);
    if (rc) return rc;
    hipStream_t st = c->st[0];
    hipLaunchKernelGGL(qzk_adler_chunks_kernel, dim3(nchunks), dim3(QZK_HT), 0, st, d_data, n, chunk_sz, nchunks, (uint32_t *)c->d_aux);
    HIPCHK(c, hipMemcpyAsync(h_adler, c->d_aux, (size_t)nchunks * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    return QZD_OK;
}

/* CRC-32 of d_data[0..n) folded on the host from 256 KiB ranges */
extern "C" int qzd_crc32(qzd_ctx *c, const uint8_t *d_data, uint64_t n, uint32_t *h_crc)
{
    /* CRC-32 of d_data[0..n): a workgroup per 256 KB, the values folded on the host with the multiplier of a full piece
     * computed once (qzd_crc32_fold).  No list of ranges goes to the device and the values come back through a small kernel,
     * not through the copy engine (round 5: the two copies and their waits were 0.5 ms of a 1.9 ms step at 4 GiB) */
    if (!c || !h_crc || (n && !d_data)) return QZD_ERR_PARAM;
    const uint32_t R = 256 * 1024;
    const uint32_t nr = (uint32_t)((n + R - 1) / R);
    *h_crc = 0;
    if (nr == 0) return QZD_OK;
    hipSetDevice(c->device);
    const size_t cb = (size_t)nr * 4;
    int rc = qzd_aux_reserve(c, cb + 64);
    if (rc) return rc;
    hipStream_t st = c->st[0];
    uint32_t *d_c = (uint32_t *)c->d_aux;
    HIPCHK(c, hipEventRecord(c->ev[0][2], st));
    hipLaunchKernelGGL(qzk_crc_chunks_kernel, dim3(nr), dim3(QZK_HT), 0, st, d_data, n, R, nr, d_c, (const uint32_t *)NULL);
    HIPCHK(c, hipEventRecord(c->ev[0][3], st));
    HIPCHK(c, ctl_copy(c->h_aux, d_c, cb, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    *h_crc = qzd_crc32_fold((const uint32_t *)c->h_aux, nr, R, n);
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev[0][2], c->ev[0][3]) == hipSuccess) c->inf_ms[1] += t;
    return QZD_OK;
}

static int find_markers(qzd_ctx *c, const uint8_t *d_src, uint64_t n, std::vector<uint32_t> &pos, hipStream_t on = NULL)
{
    pos.clear();
    if (n < 4) return QZD_OK;
    uint32_t cap = (uint32_t)(n / 256 + 4096);
    int rc = qzd_aux_reserve(c, (size_t)cap * 4 + 64);
    if (rc) return rc;
    uint32_t *d_cnt = (uint32_t *)c->d_aux, *d_list = d_cnt + 4;
    hipStream_t st = on ? on : c->st[0];
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, st));
    hipLaunchKernelGGL(qzk_marker_kernel, dim3(2048), dim3(256), 0, st, d_src, n, d_list, cap - 8, d_cnt);
    HIPCHK(c, ctl_copy(c->h_aux, d_cnt, 16, st));
    HIPCHK(c, hipStreamSynchronize(st));
    uint32_t cnt = *(uint32_t *)c->h_aux;
    if (cnt > cap - 8) return 1;       /* too many candidates: caller falls back to the serial walk */
    if (cnt) {
        HIPCHK(c, ctl_copy(c->h_aux, d_list, (size_t)cnt * 4, st));
        HIPCHK(c, hipStreamSynchronize(st));
        /* the kernel's atomics hand the positions out in no order: three 11-bit counting passes (a 2 GiB call has 32768
         * of them; std::sort took over a millisecond between two kernels) */
        const uint32_t *in = (const uint32_t *)c->h_aux;
        std::vector<uint32_t> tmp(cnt);
        pos.resize(cnt);
        for (int pass = 0; pass < 3; pass++) {
            const int sh = 11 * pass;
            uint32_t hist[2049] = {0};
            const uint32_t *src = pass == 0 ? in : pass == 1 ? tmp.data() : pos.data();
            uint32_t *dst = pass == 1 ? pos.data() : tmp.data();
            for (uint32_t i = 0; i < cnt; i++) hist[((src[i] >> sh) & 2047u) + 1]++;
            for (int k = 0; k < 2048; k++) hist[k + 1] += hist[k];
            for (uint32_t i = 0; i < cnt; i++) dst[hist[(src[i] >> sh) & 2047u]++] = src[i];
        }
        pos.swap(tmp);              /* pass 0: in -> tmp, 1: tmp -> pos, 2: pos -> tmp */
    }
    if (getenv("QATZIP_AMD_MARKER_CHECK")) {
        /* the old scan over the same bytes: the same positions, or the call fails (a missed marker would only cost speed -
         * the decode falls back to slower paths - so no parity test would see it) */
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, st));
        hipLaunchKernelGGL(qzk_marker_ref_kernel, dim3(2048), dim3(256), 0, st, d_src, n, d_list, cap - 8, d_cnt);
        HIPCHK(c, ctl_copy(c->h_aux, d_cnt, 16, st));
        HIPCHK(c, hipStreamSynchronize(st));
        const uint32_t rcnt = *(uint32_t *)c->h_aux;
        bool same = rcnt == cnt;
        if (same && rcnt) {
            HIPCHK(c, ctl_copy(c->h_aux, d_list, (size_t)rcnt * 4, st));
            HIPCHK(c, hipStreamSynchronize(st));
            std::vector<uint32_t> ref((const uint32_t *)c->h_aux, (const uint32_t *)c->h_aux + rcnt);
            std::sort(ref.begin(), ref.end());
            same = ref == pos;
        }
        if (!same) { snprintf(c->err, sizeof(c->err), "marker scan: %u positions, the reference scan finds %u or other ones", cnt, rcnt); return QZD_ERR_HIP; }
    }
    return QZD_OK;
}

/* Launch all candidate segments, ordered by compressed size: lanes (K3b) or waves of one workgroup then carry
 * segments of similar work, which bounds the divergence / tail of mixed data.  Results come back in candidate order. */
static int inflate_grouped(qzd_ctx *c, const uint8_t *d_src, uint8_t *d_dst, std::vector<qzk_infseg> &segs,
                           std::vector<qzk_infres> &res, const std::vector<uint32_t> &start, uint64_t n)
{
    const uint32_t ns = (uint32_t)segs.size();
    std::vector<uint32_t> order(ns);
    auto clen = [&](uint32_t k) { return (k + 1 < ns ? start[k + 1] : (uint32_t)n) - start[k]; };
    {   /* stable counting sort on the compressed length in 32-byte classes: grouping only needs "similar", and an exact
         * sort would seat byte-identical segments (tiled test data) in the same wave, where they never diverge */
        const uint32_t NB = 8192;
        std::vector<uint32_t> cnt(NB + 1, 0);
        auto cls = [&](uint32_t k) { uint32_t v = clen(k) >> 5; return v < NB ? v : NB - 1; };
        /* largest first: the segments with the most symbols are the critical path, so their waves start first */
        auto rcls = [&](uint32_t k) { return NB - 1 - cls(k); };
        for (uint32_t i = 0; i < ns; i++) cnt[rcls(i) + 1]++;
        for (uint32_t i = 0; i < NB; i++) cnt[i + 1] += cnt[i];
        for (uint32_t i = 0; i < ns; i++) order[cnt[rcls(i)]++] = i;
    }
    std::vector<qzk_infseg> ps(ns);
    std::vector<qzk_infres> pr(ns);
    for (uint32_t i = 0; i < ns; i++) ps[i] = segs[order[i]];
    std::vector<uint32_t> nat;                              /* segs[] is in output order: where each one went */
    if (c->so_host) { nat.resize(ns); for (uint32_t i = 0; i < ns; i++) nat[order[i]] = i; c->so_nat = nat.data(); }
    int rc = qzd_inflate_segments(c, d_src, d_dst, ps.data(), ns, pr.data());
    c->so_nat = NULL;
    if (rc) return rc;
    for (uint32_t i = 0; i < ns; i++) res[order[i]] = pr[i];
    return QZD_OK;
}

static int map_status(int st)
{
    switch (st) {
    case QZK_INF_EOUT: return QZD_ERR_DSTCAP;
    default: return QZD_ERR_DATA;
    }
}

static int inflate_stream(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                          uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc, uint8_t *h_dst, int *h_sent);

extern "C" int qzd_inflate_stream(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                                  uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc)
{
    return inflate_stream(c, d_src, n, d_dst, dst_cap, seg_hint, h_in_used, h_out_len, h_crc, NULL, NULL);
}

/* the same, for a caller that wants the output in host memory (qzDecompress): when the stream decodes in the first,
 * optimistic pass, its output is sent to h_dst range by range behind the kernels and *h_sent is set; otherwise
 * *h_sent = 0 and the caller copies d_dst out itself */
extern "C" int qzd_inflate_stream_to_host(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                                          uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc,
                                          uint8_t *h_dst, int *h_sent)
{
    if (!h_dst || !h_sent) return QZD_ERR_PARAM;
    return inflate_stream(c, d_src, n, d_dst, dst_cap, seg_hint, h_in_used, h_out_len, h_crc, h_dst, h_sent);
}

static int inflate_stream(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                          uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc, uint8_t *h_dst, int *h_sent)
{
    if (h_sent) *h_sent = 0;
    if (!c || !d_src || !h_in_used || !h_out_len) return QZD_ERR_PARAM;
    if (n == 0 || n > 0xffffffffull) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    c->inf_ms[0] = c->inf_ms[1] = c->inf_ms[2] = c->inf_ms[3] = 0;
    *h_in_used = 0; *h_out_len = 0;
    /* QATZIP_AMD_TRACE=1: wall-clock of the host-side steps on stderr (developer aid) */
    static const bool trace = getenv("QATZIP_AMD_TRACE") != NULL;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[qzd_inflate_stream] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    std::vector<uint32_t> mk;
    int rc = find_markers(c, d_src, n, mk);
    lap("marker scan + sort");
    if (rc < 0) return rc;
    const bool scan_ok = rc == 0;
    std::vector<uint32_t> start;            /* candidate segment starts */
    start.push_back(0);
    if (scan_ok) for (uint32_t p : mk) if (p < n) start.push_back(p);
    const uint32_t ns = (uint32_t)start.size();
    std::vector<qzk_infseg> segs;           /* (sized where the passes below begin: the K-lane path has lists of its own, and 3 MB of zeros are a tenth of a millisecond) */
    std::vector<qzk_infres> res;
    uint64_t total_out = 0, total_in = 0;
    bool done = false;

    /* --- 2a. thousands of candidates: phase A (Huffman decoding into position-independent token streams) runs over
     * every candidate, the host then walks the chain of real segments through the results - a candidate that is no
     * boundary (00 00 FF FF inside compressed data or a stored block) simply drops out - gives each real segment its
     * output offset, and phase B writes only those.  One pass whatever the candidates look like. --- */
    const char *force_path = getenv("QATZIP_AMD_INFLATE");
    const bool lanes = force_path ? force_path[0] == 'l' : true;   /* every candidate carries a length hint: K lanes per segment at any count */
    bool lanes_nomem = false;                                       /* the token scratch could not be had: the wave-per-segment passes below */
    {
        /* (a lone segment above 16 KB too: sixteen lanes decode 64 KB in 0.50 + 0.45 ms where the wave-per-segment kernel of
         * path 4 takes 1.85 - one thread of 64 KB qzDecompress calls 0.229 -> 0.423 Gbit/s, sixteen threads 3.6 -> 5.1;
         * QATZIP_AMD_LONE_WAVE=1 is the old way) */
        static const bool lone_lanes = getenv("QATZIP_AMD_LONE_WAVE") == NULL;
        if (scan_ok && seg_hint && (ns > 1 || (lone_lanes && seg_hint > 16384u)) && lanes) {
            auto clen = [&](uint32_t k) { return (k + 1 < ns ? start[k + 1] : (uint32_t)n) - start[k]; };
            std::vector<uint32_t> order(ns);
            {   /* largest compressed size first (256-byte classes, a class in stream order): the segments of a wave carry similar work
                 * and the longest start first.  Round 6 measured what it is worth (profiles/r6_phaseA_order.txt; 4 GiB of 64 KB segments):
                 * stream order 20.7 ms, classes of 4 KB 15.8 - 16.2, 1 KB 15.2 - 15.4, 256 B 14.9 - 15.3, 32 B 15.4; a wave's duration follows
                 * its longest segment's compressed length with a correlation of 0.54.  QATZIP_AMD_INFLATE_CLS=<log2 of the class> */
                const uint32_t NB = 8192;
                std::vector<uint32_t> cnt(NB + 1, 0);
                static const uint32_t cls_shift = getenv("QATZIP_AMD_INFLATE_CLS") ? (uint32_t)atoi(getenv("QATZIP_AMD_INFLATE_CLS")) : 8u;
                auto rcls = [&](uint32_t k) { uint32_t v = clen(k) >> cls_shift; return NB - 1 - (v < NB ? v : NB - 1); };
                for (uint32_t i = 0; i < ns; i++) cnt[rcls(i) + 1]++;
                for (uint32_t i = 0; i < NB; i++) cnt[i + 1] += cnt[i];
                for (uint32_t i = 0; i < ns; i++) order[cnt[rcls(i)]++] = i;
            }
            std::vector<qzk_infseg> ps(ns);
            std::vector<qzk_infres> pr(ns);
            std::vector<uint32_t> where(ns);                    /* candidate -> its place in the launch */
            for (uint32_t i = 0; i < ns; i++) {
                const uint32_t k = order[i];
                where[k] = i;
                ps[i].in_off = start[k]; ps[i].in_len = (uint32_t)(n - start[k]);
                ps[i].out_off = 0; ps[i].out_cap = seg_hint; ps[i].flags = 0; ps[i].pad = clen(k);
            }
            lap("segment records");
            HIPCHK(c, hipEventRecord(c->ev[0][0], c->st[0]));
            rc = two_phase(c, d_src, d_dst, ps.data(), ns, pr.data(), spec_lanes(ps.data(), ns), c->st[0], false);
            if (rc == QZD_ERR_NOMEM) { lanes_nomem = true; goto after_lanes; }        /* the passes below need no token scratch */
            if (rc) return rc;
            lap("phase A");
            std::vector<uint32_t> chain;                        /* launch indices of the real segments, in output order */
            uint64_t oo = 0; uint32_t k = 0; bool ok = true;
            for (;;) {
                const qzk_infres &r = pr[where[k]];
                if (r.status != QZK_INF_FINAL && r.status != QZK_INF_FLUSH) { ok = false; break; }
                ps[where[k]].out_off = oo; ps[where[k]].out_cap = r.out_len; ps[where[k]].flags = 0x80000000u;   /* marks a member */
                chain.push_back(where[k]);
                oo += r.out_len;
                if (r.status == QZK_INF_FINAL) { total_in = (uint64_t)start[k] + r.in_used; break; }
                const uint32_t nxt = start[k] + r.in_used;
                uint32_t j = k + 1;                                 /* the starts are sorted and nxt only grows */
                while (j < ns && start[j] < nxt) j++;
                if (j >= ns || start[j] != nxt) { ok = false; break; }
                k = j;
            }
            if (ok && oo > dst_cap) return QZD_ERR_DSTCAP;
            if (ok) {
                for (uint32_t i = 0; i < ns; i++) ps[i].flags = ps[i].flags == 0x80000000u ? 0 : QZK_INF_COUNT_ONLY;   /* the rest is skipped */
                c->so_sent = 0;
                rc = two_phase_resolve(c, d_src, d_dst, ps.data(), ns, chain.data(), (uint32_t)chain.size(), pr.data(), h_dst, c->st[0]);
                if (rc) return rc;
                for (uint32_t i : chain) if (pr[i].status < 0) { ok = false; break; }
                HIPCHK(c, hipEventRecord(c->ev[0][1], c->st[0]));
                HIPCHK(c, hipStreamSynchronize(c->st[0]));
                float t = 0;
                if (hipEventElapsedTime(&t, c->ev[0][0], c->ev[0][1]) == hipSuccess) c->inf_ms[0] += t;
                lap("phase B");
                if (ok) {
                    total_out = oo; done = true;
                    if (h_sent && h_dst && c->so_sent >= total_out) *h_sent = 1;
                }
            }
        }
    }

after_lanes:
    if (!done) { segs.resize(ns); res.resize(ns); }
    /* --- 2b. optimistic single pass (fewer candidates: one wave per segment) --- */
    if (!done && scan_ok && seg_hint && ns > 1 && (!lanes || lanes_nomem)) {
        for (uint32_t k = 0; k < ns; k++) {
            uint64_t oo = (uint64_t)k * seg_hint;
            segs[k].in_off = start[k]; segs[k].in_len = (uint32_t)(n - start[k]);
            segs[k].out_off = oo < dst_cap ? oo : dst_cap;
            segs[k].out_cap = (uint32_t)std::min<uint64_t>(seg_hint, dst_cap - segs[k].out_off);
            segs[k].flags = 0;
            segs[k].pad = (k + 1 < ns ? start[k + 1] : (uint32_t)n) - start[k];     /* compressed-length hint for phase A */
        }
        lap("segment records");
        c->so_host = h_dst; c->so_sent = 0;
        rc = inflate_grouped(c, d_src, d_dst, segs, res, start, n);
        c->so_host = NULL;
        if (rc) return rc;
        lap("inflate (optimistic)");
        bool ok = true; uint32_t k = 0;
        for (;; k++) {
            if (k >= ns) { ok = false; break; }
            const qzk_infres &r = res[k];
            if (r.status == QZK_INF_FINAL) { total_out = (uint64_t)k * seg_hint + r.out_len; total_in = (uint64_t)start[k] + r.in_used; break; }
            if (r.status != QZK_INF_FLUSH || r.out_len != seg_hint) { ok = false; break; }
            if (k + 1 >= ns || start[k + 1] != start[k] + r.in_used) { ok = false; break; }
        }
        done = ok;
        if (ok && h_sent && c->so_sent >= total_out) *h_sent = 1;       /* every range that holds output went out */
    }

    /* --- 3. two passes over the candidates --- */
    if (!done && scan_ok && ns > 1) {
        for (uint32_t k = 0; k < ns; k++) {
            segs[k].in_off = start[k]; segs[k].in_len = (uint32_t)(n - start[k]);
            segs[k].out_off = 0; segs[k].out_cap = 0xffffffffu; segs[k].flags = QZK_INF_COUNT_ONLY; segs[k].pad = 0;
        }
        rc = inflate_grouped(c, d_src, d_dst, segs, res, start, n);
        if (rc) return rc;
        std::vector<qzk_infseg> chain;
        uint64_t oo = 0; uint32_t k = 0; bool ok = true;
        for (;;) {
            const qzk_infres &r = res[k];
            if (r.status != QZK_INF_FINAL && r.status != QZK_INF_FLUSH) { ok = false; break; }
            qzk_infseg s = segs[k];
            s.flags = 0; s.out_off = oo; s.out_cap = r.out_len; s.in_len = r.in_used; s.pad = r.in_used;
            if (oo + r.out_len > dst_cap) return QZD_ERR_DSTCAP;
            chain.push_back(s);
            oo += r.out_len;
            if (r.status == QZK_INF_FINAL) { total_in = (uint64_t)start[k] + r.in_used; break; }
            uint32_t nxt = start[k] + r.in_used;
            auto it = std::lower_bound(start.begin() + k + 1, start.end(), nxt);
            if (it == start.end() || *it != nxt) { ok = false; break; }
            k = (uint32_t)(it - start.begin());
        }
        if (ok) {
            std::vector<qzk_infres> r2(chain.size());
            rc = qzd_inflate_segments(c, d_src, d_dst, chain.data(), (uint32_t)chain.size(), r2.data());
            if (rc) return rc;
            for (size_t i = 0; i < chain.size(); i++)
                if (r2[i].status < 0 || r2[i].out_len != chain[i].out_cap) { ok = false; break; }
            if (ok) { total_out = oo; done = true; }
        }
    }

    /* --- 4. one wave, straight through --- */
    if (!done) {
        qzk_infseg s; qzk_infres r;
        s.in_off = 0; s.in_len = (uint32_t)n; s.out_off = 0;
        s.out_cap = (uint32_t)std::min<uint64_t>(dst_cap, 0xffffffffu); s.flags = QZK_INF_THROUGH_FLUSH; s.pad = 0;
        rc = qzd_inflate_segments(c, d_src, d_dst, &s, 1, &r);
        if (rc) return rc;
        if (r.status != QZK_INF_FINAL) {
            snprintf(c->err, sizeof(c->err), "inflate failed: status %d after %u bytes", r.status, r.out_len);
            return map_status(r.status);
        }
        total_out = r.out_len; total_in = r.in_used;
    }
    *h_in_used = total_in; *h_out_len = total_out;
    lap("chain check / rest");
    if (h_crc) { rc = qzd_crc32(c, d_dst, total_out, h_crc); lap("crc32"); return rc; }
    return QZD_OK;
}

/* ---- a member that is still in HOST memory: decoded piece by piece while it arrives (round 4) ----
 * qzDecompress used to copy the whole source in, decode, and only then send the output out: for 2 GiB 13 ms of H2D, 12 ms
 * of phase A, and 37 ms of D2H that began when both were over.  Here the compressed bytes are cut into P pieces that go to
 * the device one behind the other; as soon as piece p has landed a helper context (its own streams, scratch and host thread)
 * scans it for flush markers, runs phase A over the candidates that END in the piece, takes the chain's state (next
 * expected segment start, output offset) from piece p - 1, walks its part of the chain, hands the state on, and runs phase
 * B with the output leaving for the host behind it.  So the first output is on its way when an eighth of the input has been
 * decoded, and the link out stays busy while the later pieces' phase A runs.  (The reference keeps its engine fed the same
 * way: requests are submitted while earlier ones are retired, src/qatzip.c:2103-2404.)
 * Anything unusual - too many candidates, a chain that does not close at a piece's end (00 00 FF FF inside compressed data
 * as a piece's last candidate), an error status, a destination too small - abandons the pieces: the input is all on the
 * device by then, and the call goes through inflate_stream() as before, which reports what is wrong. */
#define QZD_PIPE_MAX 8u
/* The pieces GROW: the output cannot leave before the first piece has been through both phases, and a launch of phase A
 * lasts as long as its slowest segment however few there are (3.8 ms for 16 MiB or 64) - so the first piece is small, 3 % of
 * the member, and every later one is as much larger as the link needs to stay busy: piece p + 1 must be decoded when piece
 * p has left.  Round 5 (profiles/r5_api_decompress_pieces.txt, 2047 MiB): two pieces cut at a third 53.5 ms, four equal ones
 * 50.1, six cut at 3 / 8 / 17 / 33 / 60 % 47.3 = 45.3 GB/s, 0.80 of the link.  (Rounds 4's finding that more pieces lose was
 * made with helpers that shared hardware queues: see stream_own_queue in qzd_device.hip.) */
#define QZD_PIPE_FIRST_PCT 3u              /* the first piece, percent of the member ... */
#define QZD_PIPE_MIN_BYTES (12u << 20)     /* ... but this many compressed bytes at least (~32 MiB of output, 512 segments) */
#define QZD_PIPE_GROW_NUM 9u               /* every piece 1.8 times the one before */
#define QZD_PIPE_GROW_DEN 5u
struct qzd_pipe {
    std::mutex m; std::condition_variable cv;
    uint32_t P;
    bool issued[QZD_PIPE_MAX];              /* main thread: the piece has landed (the host waits for each copy: a stream made to wait
                                             * for a copy's event was let go only when the copies behind it were over too) */
    bool cand_ok[QZD_PIPE_MAX]; uint32_t lastc[QZD_PIPE_MAX];       /* the last candidate at or before the piece's end */
    bool chain_ok[QZD_PIPE_MAX]; uint32_t nxt[QZD_PIPE_MAX]; uint64_t oo[QZD_PIPE_MAX];   /* the chain after the piece */
    bool final_seen; uint64_t total_in, total_out;
    std::mutex m_crc; uint32_t crc[QZD_PIPE_MAX]; uint64_t crc_len[QZD_PIPE_MAX]; bool want_crc;      /* CRC-32 of every piece's output, taken while later pieces are still on their way out */
    bool failed;
    std::chrono::steady_clock::time_point t0;
    std::vector<std::string> log;           /* QATZIP_AMD_TRACE: the pieces' steps, printed when the call is over (printing them
                                             * as they happen moved the helpers' launches: 61.6 ms without the trace, 50.8 with) */
};

static void pipe_fail(qzd_pipe *S, uint32_t p)
{
    std::lock_guard<std::mutex> g(S->m);
    S->failed = true;
    S->cand_ok[p] = true; S->chain_ok[p] = true;
    S->cv.notify_all();
}

static void pipe_piece_body(qzd_ctx *H, qzd_ctx *c, qzd_pipe *S, uint32_t p, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                            uint32_t seg_hint, const uint64_t *cut, uint8_t *h_dst);
/* a helper thread's entry: nothing may leave it as an exception (std::terminate inside a C API) - a piece that runs out of
 * host memory counts as failed and the member goes through as a whole */
static void pipe_piece(qzd_ctx *H, qzd_ctx *c, qzd_pipe *S, uint32_t p, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                       uint32_t seg_hint, const uint64_t *cut, uint8_t *h_dst)
{
    try { pipe_piece_body(H, c, S, p, d_src, n, d_dst, dst_cap, seg_hint, cut, h_dst); }
    catch (...) { pipe_fail(S, p); }
}
static void pipe_piece_body(qzd_ctx *H, qzd_ctx *c, qzd_pipe *S, uint32_t p, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                            uint32_t seg_hint, const uint64_t *cut, uint8_t *h_dst)
{
    hipSetDevice(H->device);
    static const bool trace = getenv("QATZIP_AMD_TRACE") != NULL;   /* developer aid: when each step of each piece was over */
    auto lap = [&](const char *what) {
        if (!trace) return;
        char line[160];
        snprintf(line, sizeof(line), "[pipe] piece %u %-18s at %8.3f ms", p, what,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - S->t0).count());
        std::lock_guard<std::mutex> g(S->m);
        S->log.push_back(line);
    };
    hipStream_t st = H->st[0];
    const bool last = p + 1 == S->P;
    {
        std::unique_lock<std::mutex> g(S->m);
        S->cv.wait(g, [&] { return S->issued[p] || S->failed; });
        if (S->failed) { g.unlock(); pipe_fail(S, p); return; }
    }
    /* candidates that end in (cut[p], cut[p + 1]]: the marker's four bytes may begin three bytes before the piece */
    const uint64_t lo = cut[p] >= 3 ? cut[p] - 3 : 0, hi = cut[p + 1];
    std::vector<uint32_t> mk;
    if (find_markers(H, d_src + lo, hi - lo, mk, st) != QZD_OK) { pipe_fail(S, p); return; }
    lap("landed, scanned");
    std::vector<uint32_t> start;
    uint32_t prev = 0;
    {
        std::unique_lock<std::mutex> g(S->m);
        if (p) { S->cv.wait(g, [&] { return S->cand_ok[p - 1]; }); prev = S->lastc[p - 1]; }
        start.push_back(prev);
        for (uint32_t q : mk) if (lo + q < n && lo + q > prev) start.push_back((uint32_t)(lo + q));
        S->lastc[p] = start.back(); S->cand_ok[p] = true;
        S->cv.notify_all();
        if (S->failed) { g.unlock(); pipe_fail(S, p); return; }
    }
    /* my segments: from every start to the next one; the last start is the next piece's first (the last piece's own, to n) */
    const uint32_t ns = last ? (uint32_t)start.size() : (uint32_t)start.size() - 1;
    const uint32_t arrived = (uint32_t)(last ? n : hi);
    auto clen = [&](uint32_t k) { return (k + 1 < start.size() ? start[k + 1] : (uint32_t)n) - start[k]; };
    std::vector<qzk_infseg> ps(ns);
    std::vector<qzk_infres> pr(ns);
    std::vector<uint32_t> where(ns), chain;
    if (ns) {
        std::vector<uint32_t> order(ns);
        {   /* largest compressed size first (32-byte classes), as inflate_stream() seats them */
            const uint32_t NB = 8192;
            std::vector<uint32_t> cnt(NB + 1, 0);
            auto rcls = [&](uint32_t k) { uint32_t v = clen(k) >> 5; return NB - 1 - (v < NB ? v : NB - 1); };
            for (uint32_t i = 0; i < ns; i++) cnt[rcls(i) + 1]++;
            for (uint32_t i = 0; i < NB; i++) cnt[i + 1] += cnt[i];
            for (uint32_t i = 0; i < ns; i++) order[cnt[rcls(i)]++] = i;
        }
        for (uint32_t i = 0; i < ns; i++) {
            const uint32_t k = order[i];
            where[k] = i;
            ps[i].in_off = start[k]; ps[i].in_len = arrived - start[k];     /* what has landed - not a byte more */
            ps[i].out_off = 0; ps[i].out_cap = seg_hint; ps[i].flags = 0; ps[i].pad = clen(k);
        }
        lap("phase A begins");
        if (two_phase(H, d_src, d_dst, ps.data(), ns, pr.data(), spec_lanes(ps.data(), ns), st, false) != QZD_OK) { pipe_fail(S, p); return; }
        lap("phase A");
    }
    /* the chain through my segments, from where the piece before left it */
    uint32_t want = 0; uint64_t oo = 0; bool fin = false;
    {
        std::unique_lock<std::mutex> g(S->m);
        if (p) { S->cv.wait(g, [&] { return S->chain_ok[p - 1]; }); want = S->nxt[p - 1]; oo = S->oo[p - 1]; }
        fin = S->final_seen;
        if (S->failed) { g.unlock(); pipe_fail(S, p); return; }
    }
    bool ok = true; uint64_t t_in = 0;
    const uint64_t oo0 = oo;                                        /* where my output begins */
    if (!fin) {
        uint32_t k = 0;
        while (k < start.size() && start[k] < want) k++;
        if (k >= start.size() || start[k] != want) ok = false;
        while (ok && k < ns) {
            const qzk_infres &r = pr[where[k]];
            if (r.status != QZK_INF_FINAL && r.status != QZK_INF_FLUSH) { ok = false; break; }
            ps[where[k]].out_off = oo; ps[where[k]].out_cap = r.out_len; ps[where[k]].flags = 0x80000000u;
            chain.push_back(where[k]);
            oo += r.out_len;
            if (oo > dst_cap) { ok = false; break; }
            if (r.status == QZK_INF_FINAL) { fin = true; t_in = (uint64_t)start[k] + r.in_used; break; }
            const uint32_t nx = start[k] + r.in_used;
            uint32_t j = k + 1;
            while (j < start.size() && start[j] < nx) j++;
            if (j >= start.size() || start[j] != nx) { ok = false; break; }
            k = j;
        }
        if (ok && !fin && last) ok = false;                         /* the stream ends without its final block */
    }
    {
        std::lock_guard<std::mutex> g(S->m);
        if (!ok) S->failed = true;
        S->nxt[p] = start.back(); S->oo[p] = oo; S->chain_ok[p] = true;
        if (ok && fin && !S->final_seen) { S->final_seen = true; S->total_in = t_in; S->total_out = oo; }
        S->cv.notify_all();
        if (S->failed) return;
    }
    lap("chain");
    if (chain.empty()) return;
    H->inf_ms[2] = 0;
    {   /* what phase A of this piece took, for qzd_last_inflate_timing() of the call (the pieces' kernels overlap: a sum of
         * kernel times, not a span) */
        std::lock_guard<std::mutex> g(S->m);
        c->inf_ms[3] += H->inf_ms[3]; c->inf_ms[0] += H->inf_ms[3];
        H->inf_ms[3] = 0;
    }
    for (uint32_t i = 0; i < ns; i++) ps[i].flags = ps[i].flags == 0x80000000u ? 0 : QZK_INF_COUNT_ONLY;
    /* my output's CRC-32 (the caller combines the pieces'), taken on the caller's context - idle while the pieces run - behind
     * my phase B and WHILE my output leaves: one pass over 2 GiB after the last piece had left cost the call 0.7 ms with the
     * link idle */
    uint32_t crc_mine = 0; bool crc_done = false;
    const std::function<void()> crc_fn = [&] {
        std::lock_guard<std::mutex> g(S->m_crc);
        crc_done = hipStreamWaitEvent(c->st[0], H->ev[1][2], 0) == hipSuccess && qzd_crc32(c, d_dst + oo0, oo - oo0, &crc_mine) == QZD_OK;
    };
    if (two_phase_resolve(H, d_src, d_dst, ps.data(), ns, chain.data(), (uint32_t)chain.size(), pr.data(), h_dst, st, c->pq_out,
                          S->want_crc && oo > oo0 ? &crc_fn : NULL) != QZD_OK) { pipe_fail(S, p); return; }
    for (uint32_t i : chain) if (pr[i].status < 0) { pipe_fail(S, p); return; }
    if (trace) {
        char line[160];
        snprintf(line, sizeof(line), "[pipe] piece %u phase B kernels %.3f ms, %u segments", p, H->inf_ms[2], (uint32_t)chain.size());
        std::lock_guard<std::mutex> g(S->m);
        S->log.push_back(line);
    }
    { std::lock_guard<std::mutex> g(S->m); c->inf_ms[2] += H->inf_ms[2]; c->inf_ms[0] += H->inf_ms[2]; }
    lap("phase B, sent");
    if (crc_done) { std::lock_guard<std::mutex> g(S->m); S->crc[p] = crc_mine; S->crc_len[p] = oo - oo0; }
}

/* d_src: where the n bytes at h_src are to stand on the device (they are all there when this returns, whatever it
 * returns); the rest as qzd_inflate_stream_to_host */
extern "C" int qzd_inflate_stream_from_host(qzd_ctx *c, const uint8_t *h_src, uint64_t n, uint8_t *d_src, uint8_t *d_dst, uint64_t dst_cap,
                                            uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc,
                                            uint8_t *h_dst, int *h_sent)
{
    if (!c || !h_src || !d_src || !h_dst || !h_sent || !h_in_used || !h_out_len) return QZD_ERR_PARAM;
    if (n == 0 || n > 0xffffffffull) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    *h_sent = 0;
    /* the plan: cut[0] = 0 < cut[1] < ... < cut[P] = n */
    uint64_t cut[QZD_PIPE_MAX + 1];
    uint32_t P = 0;
    {
        uint64_t len = std::max<uint64_t>(n * QZD_PIPE_FIRST_PCT / 100, QZD_PIPE_MIN_BYTES), at = 0;
        cut[0] = 0;
        while (P + 1 < QZD_PIPE_MAX && at + len + len / 2 < n) {    /* (a last piece smaller than half its predecessor joins it) */
            at = (at + len) & ~(uint64_t)4095; cut[++P] = at;
            len = len * QZD_PIPE_GROW_NUM / QZD_PIPE_GROW_DEN;
        }
        cut[++P] = n;
    }
    const char *pe = getenv("QATZIP_AMD_PIPE");                     /* pieces (0 / 1: the whole member at once); equal ones unless QATZIP_AMD_PIPE_CUTS says where */
    if (pe) {
        P = (uint32_t)std::min<int>(QZD_PIPE_MAX, std::max(0, atoi(pe)));
        for (uint32_t p = 0; p <= P; p++) cut[p] = p == P ? n : (n * p / P) & ~(uint64_t)4095;
    }
    if (!seg_hint) P = 0;
    if (P >= 2 && qzd_pipe_streams(c) != QZD_OK) { (void)hipGetLastError(); P = 0; }
    for (uint32_t p = 0; p < P; p++) {
        if (!c->pipe_ctx[p] && qzd_create_helper(c->device, &c->pipe_ctx[p]) != QZD_OK) { P = p; break; }
        if (!c->pipe_ev[p] && hipEventCreateWithFlags(&c->pipe_ev[p], hipEventDisableTiming) != hipSuccess) { P = p; break; }
    }
    if (P >= 2) cut[P] = n;                                         /* (fewer helpers than planned: the last one takes the rest) */
    if (P < 2) {
        HIPCHK(c, hipMemcpyAsync(d_src, h_src, n, hipMemcpyHostToDevice, c->st[0])); HIPCHK(c, hipStreamSynchronize(c->st[0]));      /* (not the null stream) */
        return inflate_stream(c, d_src, n, d_dst, dst_cap, seg_hint, h_in_used, h_out_len, h_crc, h_dst, h_sent);
    }
    c->inf_ms[0] = c->inf_ms[1] = c->inf_ms[2] = c->inf_ms[3] = 0;
    qzd_pipe S;
    S.t0 = std::chrono::steady_clock::now();
    S.P = P; S.final_seen = false; S.failed = false; S.total_in = S.total_out = 0;
    for (uint32_t p = 0; p < QZD_PIPE_MAX; p++) { S.issued[p] = S.cand_ok[p] = S.chain_ok[p] = false; S.lastc[p] = S.nxt[p] = 0; S.oo[p] = 0; S.crc[p] = 0; S.crc_len[p] = 0; }
    S.want_crc = h_crc != NULL;
    if (const char *ce = getenv("QATZIP_AMD_PIPE_CUTS")) {          /* developer aid: the pieces' boundaries in percent, "10,40" */
        uint32_t k = 1;
        for (const char *q = ce; *q && k < P; k++) { cut[k] = (n * (uint64_t)std::min(100, std::max(0, atoi(q))) / 100) & ~(uint64_t)4095; while (*q && *q != ',') q++; if (*q) q++; }
    }
    for (uint32_t p = 1; p <= P; p++) if (cut[p] < cut[p - 1]) cut[p] = cut[p - 1];      /* boundaries only ever go up */
    std::vector<std::thread> th;
    th.reserve(P);
    for (uint32_t p = 0; p < P; p++) {
        /* a piece whose thread cannot be started counts as failed (the others must not wait for it), and the call goes
         * through as a whole below - nothing may leave this C entry point as an exception */
        try { th.emplace_back(pipe_piece, c->pipe_ctx[p], c, &S, p, (const uint8_t *)d_src, n, d_dst, dst_cap, seg_hint, (const uint64_t *)cut, h_dst); }
        catch (...) { pipe_fail(&S, p); }
    }
    /* the copy, two pieces in flight; a piece is the helpers' when the host has seen its copy end.  (A pageable source makes
     * every copy block until it is over: the same, one at a time.) */
    bool copy_ok = true;
    auto landed = [&](uint32_t p) {
        if (copy_ok && hipEventSynchronize(c->pipe_ev[p]) != hipSuccess) copy_ok = false;
        std::lock_guard<std::mutex> g(S.m);
        if (!copy_ok) S.failed = true;
        S.issued[p] = true;
        S.cv.notify_all();
    };
    for (uint32_t p = 0; p < P; p++) {
        if (copy_ok && (hipMemcpyAsync(d_src + cut[p], h_src + cut[p], cut[p + 1] - cut[p], hipMemcpyHostToDevice, c->pq_copy) != hipSuccess ||
                        hipEventRecord(c->pipe_ev[p], c->pq_copy) != hipSuccess)) copy_ok = false;
        if (p) landed(p - 1);
    }
    landed(P - 1);
    for (auto &t : th) t.join();
    for (const std::string &l : S.log) fprintf(stderr, "%s\n", l.c_str());
    HIPCHK(c, hipStreamSynchronize(c->pq_copy));
    if (!copy_ok) {
        snprintf(c->err, sizeof(c->err), "host-to-device copy of a piece failed");
        return QZD_ERR_HIP;
    }
    if (S.failed || !S.final_seen) {
        if (getenv("QATZIP_AMD_TRACE")) fprintf(stderr, "[qzd_inflate_stream_from_host] pieces abandoned, the member goes through as a whole\n");
        return inflate_stream(c, d_src, n, d_dst, dst_cap, seg_hint, h_in_used, h_out_len, h_crc, h_dst, h_sent);
    }
    *h_in_used = S.total_in; *h_out_len = S.total_out; *h_sent = 1;
    if (h_crc) {
        uint32_t crc = 0; uint64_t have = 0;
        for (uint32_t p = 0; p < P; p++)
            if (S.crc_len[p]) { crc = have ? qzd_crc32_combine(crc, S.crc[p], S.crc_len[p]) : S.crc[p]; have += S.crc_len[p]; }
        if (have != S.total_out) return qzd_crc32(c, d_dst, S.total_out, h_crc);     /* (cannot happen: every piece on the chain reports) */
        *h_crc = crc;
    }
    return QZD_OK;
}

#ifdef QZK_SPEC_PROF
extern "C" int qzd_spec_prof(unsigned long long *out, uint32_t nwaves)
{
    static unsigned long long zero[8192][8];
    if (!out) {                                                     /* reset */
        unsigned long long st[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(qzk_stamp), st, sizeof(st), 0, hipMemcpyHostToDevice);
        hipMemcpyToSymbol(HIP_SYMBOL(qzk_stamp_b), st, 16, 0, hipMemcpyHostToDevice);
        return hipMemcpyToSymbol(HIP_SYMBOL(qzk_spec_prof), zero, sizeof(zero), 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
    }
    if (nwaves == 0xffffffffu) {
        hipMemcpyFromSymbol(out, HIP_SYMBOL(qzk_stamp), 64, 0, hipMemcpyDeviceToHost);
        return hipMemcpyFromSymbol(out + 4, HIP_SYMBOL(qzk_stamp_b), 16, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(qzk_spec_prof), (size_t)(nwaves < 8192 ? nwaves : 8192) * 64, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

#ifdef QZK_SPEC_PROF
extern __device__ unsigned int qzk_spec_seg[1 << 17];
/* profiling builds: the per-segment clocks of the last phase A (launch order), n <= 131072 */
extern "C" int qzd_spec_seg_prof(unsigned int *out, uint32_t n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(qzk_spec_seg), (size_t)(n < (1u << 17) ? n : (1u << 17)) * 4, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

/* developer aid: resident workgroups per CU the runtime grants phase A's kernels */
extern "C" int qzd_inflate_occupancy(int out[4])
{
    int a = -1, b = -1, c4 = -1, d = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, qzk_inflate_tok_kernel<16, QZD_TOK_OCC>, 16, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, qzk_inflate_spec_kernel<4, 2>, 64, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&c4, qzk_inflate_spec_kernel<8, 3>, 64, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, qzk_lz_resolve_kernel, 64 * QZK_RES_WAVES, 0);
    out[0] = a; out[1] = b; out[2] = c4; out[3] = d;
    return 0;
}

extern "C" uint64_t qzd_inflate_scratch_bytes(qzd_ctx *c) { return c ? (uint64_t)c->big_cap : 0; }

extern "C" int qzd_last_inflate_timing(qzd_ctx *c, float ms[4])
{
    if (!c || !ms) return QZD_ERR_PARAM;
    ms[0] = c->inf_ms[0]; ms[1] = c->inf_ms[1]; ms[2] = c->inf_ms[2]; ms[3] = c->inf_ms[3];
    return QZD_OK;
}
