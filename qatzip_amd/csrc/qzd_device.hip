/*
 * qzd_device.hip — host side of the device-resident C ABI (include/qzamd_device.h)
 * and the small utility kernels (size scan, slot gather).  gfx950 only.
 *
 * Level 1, input already in HBM: ONE launch of qzk_lz77_pull_kernel over all chunks of the call (persistent 16-wave
 * workgroups, one per CU, every wave pulling chunk numbers; the waves of a workgroup share the lines of one epoch-tagged
 * candidate table).  The wave that parsed a chunk (K1) also codes it (K2, qzk_huff_chunk) into the chunk's slot and folds
 * its CRC-32 along the input reads; then
 *   scan of the lengths (running total carried in HBM, no host round trip)
 *   gather of the slots into the contiguous destination.
 * Input still on the host (qzd_deflate_raw_from_host), 64 MiB and more: the same one launch, started at once and fed while
 * it runs - the host raises a watermark in pinned memory as the pieces of its copy land (qzk_wait_input) - and its waves
 * move the coded chunks to the destination themselves, in stream order (qzk_outp: no scan, no gather behind it).  Smaller
 * host calls: batches of qzd_ctx::batch_chunks chunks alternating over two streams, batch k+1's copy (third stream)
 * beside batch k's kernels.  Many small requests in one launch: per-slot lengths
 * (qzd_deflate_slots).  comp_lvl 2-9: K1b qzk_lz77_lane_kernel / the lazy kernels in place of K1, then qzk_huff_kernel and
 * qzk_crc_chunks_kernel as launches of their own (also level 1 with QATZIP_AMD_FUSE=0, round 1's pipeline).
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <atomic>

#include "qzd_internal.h"
#include "qzk_deflate_huff.h"
#include "qzk_checksum.h"

/* per call: chunk offsets (8 B each) and, behind them, what a launch that moves the stream itself needs (qzk_outp): the
 * front word and one published length per chunk */
#define QZD_OFFS_BYTES(nchunks) ((size_t)(nchunks) * 12 + 16)

/* ------------------------------------------------------------------ utility kernels */
/* offs[i] = *running + sum(len[0..i)); then *running += sum.  One 1024-thread workgroup. */
__global__ void qzk_scan_kernel(const uint32_t *len, uint32_t nchunks, uint64_t *offs, uint64_t *running)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x, per = (nchunks + 1023) / 1024;
    uint32_t b = t * per, e = b + per;
    if (b > nchunks) b = nchunks;
    if (e > nchunks) e = nchunks;
    uint64_t s = 0;
    for (uint32_t i = b; i < e; i++) s += len[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t base = *running + part[t] - s;
    for (uint32_t i = b; i < e; i++) { offs[i] = base; base += len[i]; }
    __syncthreads();
    if (t == 1023) *running += part[1023];
}

/* copy slot i (len[i] bytes) to dst + offs[i]; flag overflow instead of writing past cap */
__global__ void qzk_gather_kernel(const uint8_t *slots, uint32_t stride, const uint32_t *len, const uint64_t *offs,
                                  uint32_t nchunks, uint8_t *dst, uint64_t cap, uint32_t *overflow)
{
    const uint32_t c = blockIdx.x;
    if (c >= nchunks) return;
    const uint8_t *s = slots + (uint64_t)c * stride;
    const uint32_t n = len[c];
    const uint64_t o = offs[c];
    if (o + n > cap) { if (threadIdx.x == 0) atomicOr(overflow, 1u); return; }
    uint8_t *d = dst + o;
    const uint32_t nw = n >> 2;
    for (uint32_t i = threadIdx.x; i < nw; i += blockDim.x)
        ((qz_u32u *)d)[i].v = ((const uint32_t *)s)[i];
    for (uint32_t i = (nw << 2) + threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}

/* plain streaming copy, 16 bytes per lane and trip: the yardstick the roofline fractions are quoted against */
__global__ void __launch_bounds__(256) qzk_copy16_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {                 /* four loads in flight per lane */
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

/* ------------------------------------------------------------------ context (qzd_internal.h) */
static qzd_k1pool g_k1pool[QZD_MAX_DEVICES];
static pthread_once_t g_k1pool_once = PTHREAD_ONCE_INIT;
static void k1pool_init(void)
{
    /* QATZIP_AMD_EPOCH0: where the chunk epochs start (tests put it just below the 32-bit wrap) */
    uint32_t e0 = 1;
    if (const char *e = getenv("QATZIP_AMD_EPOCH0")) { unsigned long long v = strtoull(e, NULL, 0); if (v > 0 && v < 0xffffffffull) e0 = (uint32_t)v; }
    for (int i = 0; i < QZD_MAX_DEVICES; i++) { memset(&g_k1pool[i], 0, sizeof(g_k1pool[i])); pthread_mutex_init(&g_k1pool[i].lock, NULL); g_k1pool[i].epoch = e0; }
}

int qzd_aux_reserve(qzd_ctx *c, size_t n)
{
    if (n <= c->aux_cap) return QZD_OK;
    hipDeviceSynchronize();
    if (c->d_aux) hipFree(c->d_aux);
    if (c->h_aux) hipHostFree(c->h_aux);
    c->d_aux = NULL; c->h_aux = NULL; c->aux_cap = 0;
    n = (n + 65535) & ~(size_t)65535;
    HIPCHK(c, hipMalloc(&c->d_aux, n));
    HIPCHK(c, hipHostMalloc((void **)&c->h_aux, n, hipHostMallocDefault));
    c->aux_cap = n;
    return QZD_OK;
}

static uint32_t crc_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

uint32_t qzd_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    /* x^(2^n) mod P, filled once behind the thread-safe initialisation of a function-local static (sessions on
     * different threads make their first call concurrently) */
    struct X2N { uint32_t v[32]; X2N() { uint32_t p = 1u << 30; v[0] = p; for (int i = 1; i < 32; i++) v[i] = p = crc_multmodp(p, p); } };
    static const X2N tab;
    const uint32_t *x2n = tab.v;
    uint32_t p = 1u << 31; unsigned k = 3;
    for (uint64_t n = len2; n; n >>= 1, k++) if (n & 1) p = crc_multmodp(x2n[k & 31], p);
    return crc_multmodp(p, crc1) ^ crc2;
}

/* CRC-32 of a whole buffer from the CRC-32s of its chunks (chunk_sz bytes each, the last one n - (nchunks - 1) * chunk_sz):
 * crc32_combine folded left to right, the multiplier of a full chunk computed once */
extern "C" uint32_t qzd_crc32_fold(const uint32_t *h_crc, uint32_t nchunks, uint32_t chunk_sz, uint64_t n)
{
    if (!h_crc || nchunks == 0) return 0;
    uint32_t crc = h_crc[0];
    if (nchunks == 1) return crc;
    const uint32_t op = qzd_crc32_combine(1u << 31, 0, chunk_sz);      /* x^(8 * chunk_sz) mod P */
    for (uint32_t i = 1; i + 1 < nchunks; i++) crc = crc_multmodp(op, crc) ^ h_crc[i];
    const uint64_t tail = n - (uint64_t)(nchunks - 1) * chunk_sz;
    return qzd_crc32_combine(crc, h_crc[nchunks - 1], tail);
}

extern "C" int qzd_ctx_device(qzd_ctx *c) { return c ? c->device : -1; }

extern "C" int qzd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* A stream with a HARDWARE QUEUE OF ITS OWN.  The runtime seats ordinary streams on a pool of four hardware queues (per
 * priority), whichever has the fewest users; the packets of a queue go in order, so two streams that share one run their
 * kernels and copies one behind the other.  Which streams share depends on what the process created before: the pieces of
 * qzd_inflate_stream_from_host took 50.3 ms or 61.5 for the same 2 GiB member - a helper's marker scan waited behind all
 * 14 ms of the copy in, or behind its sibling's phase A (profiles/r5_api_decompress_pieces.txt).  A stream created with a
 * CU mask is never pooled; the mask here names every CU.  (Such a stream is a blocking one: this library puts nothing on
 * the null stream while the streams made here are busy.)  Falls back to a stream of the highest priority, then to an
 * ordinary one.  Hardware queues are few: a process holds at most QZD_OWN_QUEUES of these per device across all its
 * sessions (QATZIP_AMD_OWN_QUEUES=<n>; sixty-four sessions with two each ran the small-call sweep at a fifth of its rate) -
 * the sessions that come later get pooled streams and a slower piece-wise decode, not a slower everything. */
#define QZD_OWN_QUEUES 20
static std::atomic<int> g_own_queues[QZD_MAX_DEVICES];
static hipError_t stream_own_queue(hipStream_t *st, int device, qzd_ctx *owner)
{
    hipDeviceProp_t prop;
    int cap = QZD_OWN_QUEUES;
    const char *ce = getenv("QATZIP_AMD_OWN_QUEUES");
    if (ce) cap = atoi(ce);
    std::atomic<int> &held = g_own_queues[device % QZD_MAX_DEVICES];
    if (!getenv("QATZIP_AMD_POOLED_STREAMS") && held.load() < cap && hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        uint32_t mask[32];
        const uint32_t cus = (uint32_t)std::min(prop.multiProcessorCount, 1024), words = (cus + 31) / 32;
        for (uint32_t w = 0; w < words; w++) mask[w] = w + 1 < words || (cus & 31u) == 0 ? 0xffffffffu : (1u << (cus & 31u)) - 1u;
        if (hipExtStreamCreateWithCUMask(st, words, mask) == hipSuccess) { held++; owner->own_queues++; return hipSuccess; }
        (void)hipGetLastError();
    }
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    if (hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
/* the two copy streams of a piece-wise decode, made when the first such call comes: a process of many sessions that only
 * ever make small calls (the reference's fleet shape) must not pay for them - sixty-four sessions with two queues of their
 * own each ran the small-call sweep at a fifth of its rate (hardware queues are few; profiles/r5_small_calls.txt) */
int qzd_pipe_streams(qzd_ctx *c)
{
    if (c->pq_copy && c->pq_out) return QZD_OK;
    if (!c->pq_copy && stream_own_queue(&c->pq_copy, c->device, c) != hipSuccess) { c->pq_copy = NULL; return QZD_ERR_HIP; }
    if (!c->pq_out && stream_own_queue(&c->pq_out, c->device, c) != hipSuccess) { c->pq_out = NULL; return QZD_ERR_HIP; }
    return QZD_OK;
}
static int ctx_create(int device, qzd_ctx **out, bool helper);
extern "C" int qzd_create(int device, qzd_ctx **out) { return ctx_create(device, out, false); }
/* the context of a helper thread (a piece of qzd_inflate_stream_from_host): it launches on st[0] and nowhere else, and every
 * stream it does not create is one fewer on the hardware queues its siblings' kernels go through */
int qzd_create_helper(int device, qzd_ctx **out) { return ctx_create(device, out, true); }
static int ctx_create(int device, qzd_ctx **out, bool helper)
{
    if (!out) return QZD_ERR_PARAM;
    *out = NULL;
    if (hipSetDevice(device) != hipSuccess) return QZD_ERR_HIP;
    {
        /* QATZIP_AMD_SYNC=block: a thread that waits for the device sleeps instead of spinning.  For hosts with fewer cores
         * (or a smaller CPU quota) than processes: the reference's fleet shape, P pinned processes of synchronous calls -
         * tools/fleet.sh, profiles/r5_fleet.txt */
        static const char *sy = getenv("QATZIP_AMD_SYNC");
        if (sy && sy[0] == 'b') { if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError(); }
    }
    qzd_ctx *c = new (std::nothrow) qzd_ctx();
    if (!c) return QZD_ERR_HIP;
    memset(c, 0, sizeof(*c));
    c->device = device; c->helper = helper;
#define QZD_CREATE_FAIL do { qzd_destroy(c); return QZD_ERR_HIP; } while (0)   /* no half-built context leaks */
    for (int i = 0; i < QZD_NBUF; i++) {
        if (helper && i > 0) c->st[i] = c->st[0];
        else if ((helper ? stream_own_queue(&c->st[i], device, c) : hipStreamCreateWithFlags(&c->st[i], hipStreamNonBlocking)) != hipSuccess) QZD_CREATE_FAIL;
        hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->k1done[i], hipEventDisableTiming);
        for (int k = 0; k < 4; k++) hipEventCreate(&c->ev[i][k]);
    }
    hipEventCreate(&c->ev_begin); hipEventCreate(&c->ev_end);
    if (helper) c->st_copy = c->st_out = c->st[0];
    else if (hipStreamCreateWithFlags(&c->st_copy, hipStreamNonBlocking) != hipSuccess ||
             hipStreamCreateWithFlags(&c->st_out, hipStreamNonBlocking) != hipSuccess) QZD_CREATE_FAIL;
    for (int i = 0; i < QZD_NBUF + 1; i++) hipEventCreateWithFlags(&c->cp_ev[i], hipEventDisableTiming);
    for (int i = 0; i < 8; i++) hipEventCreateWithFlags(&c->so_ev[i], hipEventDisableTiming);
    c->so_host = NULL; c->so_nat = NULL; c->so_sent = 0;
    for (int i = 0; i < QZD_K1EV; i++) { hipEventCreate(&c->k1ev[i][0]); hipEventCreate(&c->k1ev[i][1]); }
    {
        /* K1 residency (measured): QZD_K1_WGS_PER_CU pulling waves per CU (one workgroup), each with its
         * column of the workgroup's candidate table.  QATZIP_AMD_K1_WGS=<n> overrides the total number of waves. */
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) QZD_CREATE_FAIL;
        const uint32_t cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
        c->cus = cus;
        c->k1_wgs = QZD_K1_WGS_PER_CU * cus;
        const char *e = getenv("QATZIP_AMD_K1_WGS");
        unsigned a = 0;
        if (e && sscanf(e, "%u", &a) == 1 && a > 0 && a <= 65536) c->k1_wgs = a;
        c->batch_chunks = QZD_BATCH_ROUNDS * c->k1_wgs;
        /* the tables (4 MiB per workgroup, 5 GiB for a full device) belong to the device (g_k1pool): allocated by the
         * first call that needs them and only as many as its chunks can occupy, shared by every context on the GPU */
        if (hipMalloc(&c->k1_counter, QZD_NBUF * 4) != hipSuccess) QZD_CREATE_FAIL;
    }
    if (hipMalloc(&c->d_running, 8) != hipSuccess || hipMalloc(&c->d_overflow, 4) != hipSuccess) QZD_CREATE_FAIL;
    hipHostMalloc((void **)&c->h_running, 8, hipHostMallocDefault);
    hipHostMalloc((void **)&c->h_overflow, 4, hipHostMallocDefault);
    /* watermark of a launch whose input is still arriving: read by the kernel over the link, so coherent + mapped */
    if (hipHostMalloc((void **)&c->h_wm, 8, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) c->h_wm = NULL;
#undef QZD_CREATE_FAIL
    *out = c;
    return QZD_OK;
}

extern "C" void qzd_destroy(qzd_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (int i = 0; i < QZD_NBUF; i++) {
        hipFree(c->slots[i]);
        /* handles may be missing: qzd_create comes here from its failure paths */
        if (c->st[i] && !(c->helper && i > 0)) hipStreamDestroy(c->st[i]);
        if (c->done[i]) hipEventDestroy(c->done[i]);
        if (c->k1done[i]) hipEventDestroy(c->k1done[i]);
        if (i == 0) {
            for (int k = 0; k < 8; k++) if (c->so_ev[k]) hipEventDestroy(c->so_ev[k]);
            if (c->st_copy && !c->helper) hipStreamDestroy(c->st_copy);
            if (c->st_out && !c->helper) hipStreamDestroy(c->st_out);
            if (c->pq_copy) hipStreamDestroy(c->pq_copy);
            if (c->pq_out) hipStreamDestroy(c->pq_out);
            for (int k = 0; k < QZD_NBUF + 1; k++) if (c->cp_ev[k]) hipEventDestroy(c->cp_ev[k]);
        }
        for (int k = 0; k < 4; k++) if (c->ev[i][k]) hipEventDestroy(c->ev[i][k]);
    }
    if (c->ev_begin) hipEventDestroy(c->ev_begin);
    if (c->ev_end) hipEventDestroy(c->ev_end);
    for (int i = 0; i < QZD_K1EV; i++) { if (c->k1ev[i][0]) hipEventDestroy(c->k1ev[i][0]); if (c->k1ev[i][1]) hipEventDestroy(c->k1ev[i][1]); }
    hipFree(c->k1_counter);
    hipFree(c->d_len); hipFree(c->d_crc); hipFree(c->d_offs); hipFree(c->d_running); hipFree(c->d_overflow);
    hipHostFree(c->h_running); hipHostFree(c->h_overflow); if (c->h_wm) hipHostFree(c->h_wm);
    if (c->d_lz4tab) hipFree(c->d_lz4tab);
    if (c->d_aux) hipFree(c->d_aux);
    if (c->h_aux) hipHostFree(c->h_aux);
    if (c->d_big) hipFree(c->d_big);
    if (c->d_lane) hipFree(c->d_lane);
    if (c->d_cdesc) hipFree(c->d_cdesc);
    for (int i = 0; i < 8; i++) { if (c->pipe_ctx[i]) qzd_destroy(c->pipe_ctx[i]); if (c->pipe_ev[i]) hipEventDestroy(c->pipe_ev[i]); }
    g_own_queues[c->device % QZD_MAX_DEVICES] -= (int)c->own_queues;
    delete c;
}

/* measured HBM stream-copy rate of this device (read + written bytes per second, decimal GB): `bytes` per buffer (take
 * it well above the 256 MiB Infinity Cache), best of `iters` passes */
extern "C" int qzd_stream_copy_peak(qzd_ctx *c, uint64_t bytes, int iters, double *gbps)
{
    if (!c || !gbps || bytes < (1u << 20) || iters < 1) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    uint4 *a = NULL, *b = NULL;
    if (hipMalloc(&a, bytes) != hipSuccess) return QZD_ERR_HIP;
    if (hipMalloc(&b, bytes) != hipSuccess) { hipFree(a); return QZD_ERR_HIP; }
    hipMemsetAsync(a, 0x5a, bytes, c->st[0]);
    const size_t n16 = bytes / 16;
    float best = 0;
    for (int it = 0; it < iters + 1; it++) {                    /* pass 0 warms up */
        hipEventRecord(c->ev_begin, c->st[0]);
        hipLaunchKernelGGL(qzk_copy16_kernel, dim3(256 * 8), dim3(256), 0, c->st[0], a, b, n16);
        hipEventRecord(c->ev_end, c->st[0]);
        hipStreamSynchronize(c->st[0]);
        float t = 0;
        if (hipEventElapsedTime(&t, c->ev_begin, c->ev_end) != hipSuccess) t = 0;
        if (it > 0 && t > 0 && (best == 0 || t < best)) best = t;
    }
    hipFree(a); hipFree(b);
    if (best <= 0) return QZD_ERR_HIP;
    *gbps = 2.0 * (double)bytes / (best * 1e-3) / 1e9;
    return QZD_OK;
}

/* what the host link delivers to this device: pinned hipMemcpyAsync of `bytes`, host -> device and device -> host, best of
 * `iters`, decimal GB/s each (the bound of the host-to-host API beside the kernels; the QAT path's DMA, src/qatzip.c:1542) */
extern "C" int qzd_pcie_peak(qzd_ctx *c, uint64_t bytes, int iters, double *h2d_gbps, double *d2h_gbps)
{
    if (!c || !bytes || iters < 1) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    void *h = NULL, *d = NULL;
    if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) return QZD_ERR_HIP;
    if (hipMalloc(&d, bytes) != hipSuccess) { hipHostFree(h); return QZD_ERR_HIP; }
    memset(h, 0x5a, bytes);
    float best[2] = {0, 0};
    for (int dir = 0; dir < 2; dir++)
        for (int it = 0; it <= iters; it++) {
            hipEventRecord(c->ev_begin, c->st[0]);
            if (dir == 0) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->st[0]);
            else hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->st[0]);
            hipEventRecord(c->ev_end, c->st[0]);
            hipStreamSynchronize(c->st[0]);
            float t = 0;
            if (hipEventElapsedTime(&t, c->ev_begin, c->ev_end) != hipSuccess) t = 0;
            if (it > 0 && t > 0 && (best[dir] == 0 || t < best[dir])) best[dir] = t;
        }
    hipFree(d); hipHostFree(h);
    if (best[0] <= 0 || best[1] <= 0) return QZD_ERR_HIP;
    if (h2d_gbps) *h2d_gbps = (double)bytes / (best[0] * 1e-3) / 1e9;
    if (d2h_gbps) *d2h_gbps = (double)bytes / (best[1] * 1e-3) / 1e9;
    return QZD_OK;
}

extern "C" uint32_t qzd_batch_chunks(qzd_ctx *c) { return c ? c->batch_chunks : 0; }

extern "C" const char *qzd_last_error(qzd_ctx *c) { return c ? c->err : "no context"; }

extern "C" void *qzd_dev_alloc(qzd_ctx *c, size_t n)
{
    void *p = NULL;
    if (c) hipSetDevice(c->device);
    if (hipMalloc(&p, n ? n : 1) != hipSuccess) return NULL;
    return p;
}
extern "C" void qzd_dev_free(qzd_ctx *c, void *p) { if (c) hipSetDevice(c->device); if (p) hipFree(p); }
extern "C" int qzd_h2d(qzd_ctx *c, void *d, const void *h, size_t n)
{
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    return QZD_OK;
}
extern "C" int qzd_d2h(qzd_ctx *c, void *h, const void *d, size_t n)
{
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    return QZD_OK;
}
/* device to device, within the context's GPU (or from a peer's memory it can reach); returns when the copy is done */
extern "C" int qzd_d2d(qzd_ctx *c, void *d_dst, const void *d_src, size_t n)
{
    if (!c || (n && (!d_dst || !d_src))) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(d_dst, d_src, n, hipMemcpyDeviceToDevice, c->st[1]));
    HIPCHK(c, hipStreamSynchronize(c->st[1]));
    return QZD_OK;
}
extern "C" void *qzd_host_alloc_pinned(size_t n)
{
    void *p = NULL;
    if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return NULL;
    return p;
}
/* follow_policy: place the pages by the calling thread's NUMA memory policy (qzMalloc's node argument) */
extern "C" void *qzd_host_alloc_pinned_numa(size_t n, int follow_policy)
{
    void *p = NULL;
    if (hipHostMalloc(&p, n ? n : 1, follow_policy ? hipHostMallocNumaUser : hipHostMallocDefault) != hipSuccess) {
        if (!follow_policy || hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess) return NULL;
    }
    return p;
}
extern "C" void qzd_host_free_pinned(void *p) { if (p) hipHostFree(p); }

static uint32_t slot_stride_for(uint32_t chunk_sz) { return (chunk_sz / 8u * 9u + 1024u + 15u) & ~15u; }

/* K2 inside the K1 waves (the default) or as a launch of its own (QATZIP_AMD_FUSE=0, round 1's pipeline, kept for
 * comparison): decides the launch shape AND the layout of the device pool's scratch, so it is fixed per process */
static bool k1_fused(void)
{
    static const bool fuse = !(getenv("QATZIP_AMD_FUSE") && getenv("QATZIP_AMD_FUSE")[0] == '0');
    return fuse;
}

/* the device pool's scratch (called with the pool's lock held) and this context's per-call arrays.
 *   fused:    symbols per WAVE (a wave codes the chunk it parsed before it pulls the next: 3 bytes per input byte for the
 *             4096 resident waves, whatever the call's size), slots and meta per chunk of the CALL, one set;
 *   separate: symbols, slots and meta per chunk of a BATCH, double-buffered (K2 of batch b beside K1 of batch b+1). */
static int ensure_scratch(qzd_ctx *c, qzd_k1pool *pool, uint32_t chunk_sz, uint32_t nchunks)
{
    const bool fuse = k1_fused();
    uint32_t batch = fuse ? nchunks : (nchunks < c->batch_chunks ? nchunks : c->batch_chunks);
    const uint32_t waves = ((c->k1_wgs + QZK_K1_WAVES - 1) / QZK_K1_WAVES) * QZK_K1_WAVES;
    size_t sym = (size_t)(fuse ? waves : batch) * chunk_sz + 256, slot = (size_t)batch * slot_stride_for(chunk_sz);
    if (sym > pool->sym_cap || slot > pool->slot_cap || batch > pool->meta_cap) {
        hipDeviceSynchronize();
        sym = std::max(sym, pool->sym_cap); slot = std::max(slot, pool->slot_cap); batch = std::max(batch, pool->meta_cap);
        pool->sym_cap = 0; pool->slot_cap = 0; pool->meta_cap = 0;      /* set again only when every allocation below succeeded */
        for (int i = 0; i < QZD_NBUF; i++) {
            hipFree(pool->sym_lc[i]); hipFree(pool->sym_dist[i]); hipFree(pool->slots[i]); hipFree(pool->meta[i]);
            pool->sym_lc[i] = NULL; pool->sym_dist[i] = NULL; pool->slots[i] = NULL; pool->meta[i] = NULL;
        }
        for (int i = 0; i < (fuse ? 1 : QZD_NBUF); i++) {
            HIPCHK(c, hipMalloc(&pool->sym_lc[i], sym));
            HIPCHK(c, hipMalloc(&pool->sym_dist[i], sym * 2));
            HIPCHK(c, hipMalloc(&pool->slots[i], slot));
            HIPCHK(c, hipMalloc(&pool->meta[i], (size_t)batch * sizeof(qzk_lzmeta)));
        }
        pool->sym_cap = sym; pool->slot_cap = slot; pool->meta_cap = batch;
    }
    if (nchunks > c->call_cap) {
        hipDeviceSynchronize();
        hipFree(c->d_len); hipFree(c->d_crc); hipFree(c->d_offs);
        c->d_len = NULL; c->d_crc = NULL; c->d_offs = NULL; c->call_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_len, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_crc, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_offs, QZD_OFFS_BYTES(nchunks)));
        c->call_cap = nchunks;
    }
    return QZD_OK;
}

#include "qzk_deflate_lz77_lane.h"
/* K1b is zlib's loop with zlib's tables, one chunk per lane: the path of comp_lvl 2-9 (level 1 has its own kernel;
 * QATZIP_AMD_DEFLATE=lane sends level 1 here too, for the parity tests).  One batch = the whole call: K1b over every
 * chunk, then K2, scan, gather. */
static int deflate_lane_path(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level, int last,
                             uint8_t *d_dst, uint64_t dst_cap, uint32_t nchunks, const uint32_t *cdesc)
{
    const uint32_t stride = slot_stride_for(chunk_sz);
    const size_t symb = ((size_t)nchunks * chunk_sz + 511) & ~(size_t)255;
    const size_t metab = ((size_t)nchunks * sizeof(qzk_lzmeta) + 255) & ~(size_t)255;
    const size_t slotb = (size_t)nchunks * stride;
    const size_t headb = (size_t)nchunks * QZK_HSIZE * 2, prevb = (size_t)nchunks * QZK_WSIZE * 2;
    const size_t need = symb * 3 + metab + slotb + headb + prevb;
    if (need > c->lane_cap) {
        hipDeviceSynchronize();
        if (c->d_lane) hipFree(c->d_lane);
        c->d_lane = NULL; c->lane_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_lane, need));
        c->lane_cap = need;
    }
    if (nchunks > c->call_cap) {
        hipDeviceSynchronize();
        hipFree(c->d_len); hipFree(c->d_crc); hipFree(c->d_offs);
        c->d_len = NULL; c->d_crc = NULL; c->d_offs = NULL; c->call_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_len, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_crc, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_offs, QZD_OFFS_BYTES(nchunks)));
        c->call_cap = nchunks;
    }
    uint8_t *pb = c->d_lane;
    uint8_t *sym_lc = pb; pb += symb;
    uint16_t *sym_dist = (uint16_t *)pb; pb += 2 * symb;
    qzk_lzmeta *meta = (qzk_lzmeta *)pb; pb += metab;
    uint8_t *slots = pb; pb += slotb;
    uint16_t *head = (uint16_t *)pb; pb += headb;
    uint16_t *prev = (uint16_t *)pb;
    hipStream_t st = c->st[0];
    c->last_nchunks = nchunks; c->nbatches = 1;
    HIPCHK(c, hipMemsetAsync(c->d_running, 0, 8, st));
    HIPCHK(c, hipMemsetAsync(c->d_overflow, 0, 4, st));
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    HIPCHK(c, hipMemsetAsync(head, 0, headb, st));
    HIPCHK(c, hipEventRecord(c->ev[0][0], st));
    /* chunks per wave: a lane's loop is a chain of dependent HBM round trips and a wave steps at the pace of its
     * slowest lane, so the fewer lanes share a wave the better - measured 1 GiB, level 6: 64 lanes 0.23 GB/s, 16: 0.33,
     * 4: 0.53, 1: 0.69 (level 2: 2.07 -> 4.29).  One chunk per wave it is (QATZIP_AMD_LANE_LPW overrides); the idle
     * lanes are where a parallel candidate compare would go. */
    uint32_t lpw = 1;
    if (const char *e = getenv("QATZIP_AMD_LANE_LPW")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v <= 64 && !(v & (v - 1))) lpw = v; }
    hipLaunchKernelGGL(qzk_lz77_lane_kernel, dim3((nchunks + lpw - 1) / lpw), dim3(lpw), 0, st, d_src, n, chunk_sz, nchunks,
                       sym_lc, sym_dist, meta, head, prev, qzk_level_cfg(level), cdesc);
    HIPCHK(c, hipEventRecord(c->ev[0][1], st));
    hipLaunchKernelGGL(qzk_huff_kernel, dim3(nchunks), dim3(QZK_HW), 0, st, d_src, n, chunk_sz, nchunks, sym_lc, sym_dist,
                       meta, slots, stride, last ? nchunks - 1 : ~0u, c->d_len, cdesc);
    hipLaunchKernelGGL(qzk_crc_chunks_kernel, dim3(nchunks), dim3(QZK_HT), 0, st, d_src, n, chunk_sz, nchunks, c->d_crc, cdesc);
    HIPCHK(c, hipEventRecord(c->ev[0][2], st));
    hipLaunchKernelGGL(qzk_scan_kernel, dim3(1), dim3(1024), 0, st, c->d_len, nchunks, c->d_offs, c->d_running);
    hipLaunchKernelGGL(qzk_gather_kernel, dim3(nchunks), dim3(256), 0, st, slots, stride, c->d_len, c->d_offs, nchunks,
                       d_dst, dst_cap, c->d_overflow);
    HIPCHK(c, hipEventRecord(c->ev[0][3], st));
    HIPCHK(c, hipMemcpyAsync(c->h_running, c->d_running, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(c->h_overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipGetLastError());
    return QZD_OK;
}

static int deflate_lazy_path(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level, int last,
                             uint8_t *d_dst, uint64_t dst_cap, uint32_t nchunks, const uint32_t *cdesc);

#include "qzk_deflate_wide.h"
static uint64_t *g_wide_prof; static uint32_t g_wide_prof_cap, g_wide_prof_n;     /* QATZIP_AMD_WIDE_PROF=1: phase clocks of the last call */
/* Level 1, chunks of at most 64 KB, launches of at most one chunk per CU (or QATZIP_AMD_K1=wide): K1w - one chunk per
 * 1024-thread workgroup, everything it looks up on chip (qzk_deflate_wide.h) - in place of K1; then K2 / CRC / scan /
 * gather as launches of their own.  One batch = the whole call, on one stream. */
static int deflate_wide_path(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int last,
                             uint8_t *d_dst, uint64_t dst_cap, uint32_t nchunks, const uint32_t *cdesc)
{
    const uint32_t stride = slot_stride_for(chunk_sz);
    const uint32_t wgs = nchunks < c->cus ? nchunks : c->cus;
    const size_t symb = ((size_t)nchunks * chunk_sz + 511) & ~(size_t)255;
    const size_t metab = ((size_t)nchunks * sizeof(qzk_lzmeta) + 255) & ~(size_t)255;
    const size_t slotb = ((size_t)nchunks * stride + 255) & ~(size_t)255;
    const size_t prevb = (size_t)wgs * 65536 * 2;
    const size_t need = symb * 3 + metab + slotb + prevb + 256;
    if (need > c->lane_cap) {
        hipDeviceSynchronize();
        if (c->d_lane) hipFree(c->d_lane);
        c->d_lane = NULL; c->lane_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_lane, need));
        c->lane_cap = need;
    }
    if (nchunks > c->call_cap) {
        hipDeviceSynchronize();
        hipFree(c->d_len); hipFree(c->d_crc); hipFree(c->d_offs);
        c->d_len = NULL; c->d_crc = NULL; c->d_offs = NULL; c->call_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_len, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_crc, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_offs, QZD_OFFS_BYTES(nchunks)));
        c->call_cap = nchunks;
    }
    uint8_t *pb = c->d_lane;
    uint8_t *sym_lc = pb; pb += symb;
    uint16_t *sym_dist = (uint16_t *)pb; pb += 2 * symb;
    qzk_lzmeta *meta = (qzk_lzmeta *)pb; pb += metab;
    uint8_t *slots = pb; pb += slotb;
    uint16_t *prevtab = (uint16_t *)pb;
    uint64_t *xprof = NULL;
    if (const char *e = getenv("QATZIP_AMD_WIDE_PROF")) if (e[0] == '1') {
        if (nchunks > g_wide_prof_cap) { hipDeviceSynchronize(); if (g_wide_prof) hipFree(g_wide_prof); g_wide_prof = NULL; g_wide_prof_cap = 0;
            if (hipMalloc(&g_wide_prof, (size_t)nchunks * 128) == hipSuccess) g_wide_prof_cap = nchunks; }
        if (g_wide_prof) { xprof = g_wide_prof; g_wide_prof_n = nchunks; hipMemsetAsync(xprof, 0, (size_t)nchunks * 128, c->st[0]); }
    }
    hipStream_t st = c->st[0];
    c->last_nchunks = nchunks; c->nbatches = 1; c->k1ev_n = 0;
    HIPCHK(c, hipMemsetAsync(c->d_running, 0, 8, st));
    HIPCHK(c, hipMemsetAsync(c->d_overflow, 0, 4, st));
    HIPCHK(c, hipMemsetAsync(c->k1_counter, 0, 4, st));
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    HIPCHK(c, hipEventRecord(c->ev[0][0], st));
    HIPCHK(c, hipEventRecord(c->k1ev[0][0], st));
    hipLaunchKernelGGL(qzk_lz77_wide_kernel, dim3(wgs), dim3(QZX_W), 0, st, d_src, n, chunk_sz, nchunks, sym_lc, sym_dist, meta,
                       prevtab, c->k1_counter, cdesc, xprof);
    HIPCHK(c, hipEventRecord(c->k1ev[0][1], st)); c->k1ev_chunks[0] = nchunks; c->k1ev_n = 1;
    HIPCHK(c, hipEventRecord(c->ev[0][1], st));
    hipLaunchKernelGGL(qzk_huff_kernel, dim3(nchunks), dim3(QZK_HW), 0, st, d_src, n, chunk_sz, nchunks, sym_lc, sym_dist,
                       meta, slots, stride, last ? nchunks - 1 : ~0u, c->d_len, cdesc);
    hipLaunchKernelGGL(qzk_crc_chunks_kernel, dim3(nchunks), dim3(QZK_HT), 0, st, d_src, n, chunk_sz, nchunks, c->d_crc, cdesc);
    HIPCHK(c, hipEventRecord(c->ev[0][2], st));
    hipLaunchKernelGGL(qzk_scan_kernel, dim3(1), dim3(1024), 0, st, c->d_len, nchunks, c->d_offs, c->d_running);
    hipLaunchKernelGGL(qzk_gather_kernel, dim3(nchunks), dim3(256), 0, st, slots, stride, c->d_len, c->d_offs, nchunks,
                       d_dst, dst_cap, c->d_overflow);
    HIPCHK(c, hipEventRecord(c->ev[0][3], st));
    HIPCHK(c, hipMemcpyAsync(c->h_running, c->d_running, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(c->h_overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipGetLastError());
    return QZD_OK;
}

/* cdesc (device memory, or NULL): per-chunk length / closes-its-stream flag of a coalesced launch (qzk_chunk_len) */
static int deflate_enqueue_impl(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level,
                                int last, uint8_t *d_dst, uint64_t dst_cap, const uint32_t *cdesc, const uint8_t *h_src);
static int deflate_enqueue(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level,
                           int last, uint8_t *d_dst, uint64_t dst_cap, const uint32_t *cdesc, const uint8_t *h_src = NULL)
{
    return deflate_enqueue_impl(c, d_src, n, chunk_sz, level, last, d_dst, dst_cap, cdesc, h_src);
}
static int deflate_enqueue_impl(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level,
                                int last, uint8_t *d_dst, uint64_t dst_cap, const uint32_t *cdesc, const uint8_t *h_src)
{
    if (!c || !d_dst || (n && !d_src)) return QZD_ERR_PARAM;
    if (chunk_sz < 1024 || chunk_sz > 512 * 1024 || (chunk_sz & (chunk_sz - 1))) return QZD_ERR_PARAM;
    if (level < 1 || level > 9) { snprintf(c->err, sizeof(c->err), "deflate level %d: zlib has levels 1-9", level); return QZD_ERR_UNSUPPORTED; }
    if (n > ((uint64_t)1 << 32)) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    const uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    {
        /* level 1: one chunk per wave (K1, window speculation over the four-newest table); levels 2-9: zlib's own loop
         * and tables, one chunk per LANE (K1b).  QATZIP_AMD_DEFLATE=lane takes level 1 through K1b as well. */
        const char *force = getenv("QATZIP_AMD_DEFLATE");
        if ((level != 1 || (force && force[0] == 'l')) && h_src && n) { HIPCHK(c, hipMemcpyAsync((void *)d_src, h_src, n, hipMemcpyHostToDevice, c->st[0])); HIPCHK(c, hipStreamSynchronize(c->st[0])); }    /* (not the null stream: it would wait for every blocking stream of the process) */
        /* the lazy levels have their own, parallel, path; QATZIP_AMD_LAZY=0 sends them through K1b instead */
        const char *lz = getenv("QATZIP_AMD_LAZY");
        if (level >= 4 && !(lz && lz[0] == '0')) return deflate_lazy_path(c, d_src, n, chunk_sz, level, last, d_dst, dst_cap, nchunks, cdesc);
        if (level != 1 || (force && force[0] == 'l')) return deflate_lane_path(c, d_src, n, chunk_sz, level, last, d_dst, dst_cap, nchunks, cdesc);
        /* Two parse kernels for level 1.  A call that fills the chip gives every wave a chunk of its own (K1,
         * qzk_lz77_pull_kernel: 4096 chunks in flight hide each other's latency; throughput).  A launch of at most one chunk
         * per CU - a lone qzCompress of a block or two, the requests a few threads have in flight - gives every chunk a
         * whole CU instead (K1w, qzk_lz77_wide_kernel: the chunk, prev[] and the window's work on chip; latency: a chunk
         * takes about half the time one wave needs).  QATZIP_AMD_K1=wide / pull forces one of them. */
        const char *k1 = getenv("QATZIP_AMD_K1");
        const bool force_wide = k1 && k1[0] == 'w', force_pull = k1 && k1[0] == 'p';
        if (chunk_sz <= 65536 && (force_wide || (!force_pull && nchunks <= c->cus + c->cus / 2))) {      /* measured crossover: profiles/r4_k1_crossover.txt */
            if (h_src && n) { HIPCHK(c, hipMemcpyAsync((void *)d_src, h_src, n, hipMemcpyHostToDevice, c->st[0])); HIPCHK(c, hipStreamSynchronize(c->st[0])); }
            return deflate_wide_path(c, d_src, n, chunk_sz, last, d_dst, dst_cap, nchunks, cdesc);
        }
    }
    const uint32_t max_wgs = (c->k1_wgs + QZK_K1_WAVES - 1) / QZK_K1_WAVES;      /* workgroups of a full launch (QZK_K1_OCC per CU) */
    qzd_k1pool *const pool = &g_k1pool[c->device % QZD_MAX_DEVICES];
    pthread_once(&g_k1pool_once, k1pool_init);
    /* The device's tables and batch scratch are used by one call at a time - on the GPU: this call's streams wait for the
     * event the previous call (of whichever context) recorded behind its last launch, and leave theirs when everything is
     * enqueued.  The mutex is held while this function runs and never across API calls. */
    struct PoolGuard {
        qzd_k1pool *p; qzd_ctx *c; bool launched;
        PoolGuard(qzd_k1pool *p_, qzd_ctx *c_) : p(p_), c(c_), launched(false) { pthread_mutex_lock(&p->lock); }
        ~PoolGuard()
        {
            if (launched) {
                /* behind everything this call put on its streams (st[1] and the copy stream are joined into st[0] on the
                 * success path; on an error path both are waited for here) */
                if (!p->busy) hipEventCreateWithFlags(&p->busy, hipEventDisableTiming);
                hipEvent_t j = c->done[1];
                if (hipEventRecord(j, c->st[1]) == hipSuccess) hipStreamWaitEvent(c->st[0], j, 0);
                if (p->busy && hipEventRecord(p->busy, c->st[0]) == hipSuccess) p->busy_valid = true;
                else { hipStreamSynchronize(c->st[0]); hipStreamSynchronize(c->st[1]); p->busy_valid = false; }
            }
            pthread_mutex_unlock(&p->lock);
        }
    } guard(pool, c);
    if (pool->busy_valid) {
        HIPCHK(c, hipStreamWaitEvent(c->st[0], pool->busy, 0));
        HIPCHK(c, hipStreamWaitEvent(c->st[1], pool->busy, 0));
    }
    {
        /* a launch of few chunks spreads them over workgroups (CUs) before it stacks them on the waves of one */
        const uint32_t want = nchunks < max_wgs ? nchunks : max_wgs;
        /* the device's tables and batch scratch: taken for the whole call (released by qzd_sync, or by the caller of
         * this function when it fails) */
        const int rc = ensure_scratch(c, pool, chunk_sz, nchunks);
        if (rc) return rc;
        if (want > pool->tab_wgs) {
            hipDeviceSynchronize();
            if (pool->tables) hipFree(pool->tables);
            pool->tables = NULL; pool->tab_wgs = 0;
            const uint32_t get = want > max_wgs / 4 ? max_wgs : want;     /* a big call: take the whole set at once */
            const size_t tb = (size_t)QZK_K1_TABROWS(get) * QZK_HSIZE * QZK_K1_TABW * sizeof(qzk_bkt);
            if (hipMalloc(&pool->tables, tb) != hipSuccess || hipMemset(pool->tables, 0, tb) != hipSuccess ||   /* epoch 0 = never valid */
                hipDeviceSynchronize() != hipSuccess) {                   /* hipMemset of device memory returns early, and the
                                                                           * (non-blocking) work streams do not wait for it */
                if (pool->tables) hipFree(pool->tables);
                pool->tables = NULL;
                snprintf(c->err, sizeof(c->err), "K1 tables: out of device memory");
                return QZD_ERR_HIP;
            }
            pool->tab_wgs = get;
        }
        if ((uint64_t)pool->epoch + nchunks + 1 >= 0xffffffffull) {        /* epochs wrapped: forget everything once */
            hipDeviceSynchronize();
            hipMemset(pool->tables, 0, (size_t)QZK_K1_TABROWS(pool->tab_wgs) * QZK_HSIZE * QZK_K1_TABW * sizeof(qzk_bkt));
            hipDeviceSynchronize();
            pool->epoch = 1;
        }
    }
    const uint32_t stride = slot_stride_for(chunk_sz);
    c->last_nchunks = nchunks;
    /* input already in HBM and K2 inside K1: the whole call is ONE launch - every launch ends with a tail (the waves
     * finish their last chunks, ~8 ms each, at different times; on the bench data ~3 ms of a 12288-chunk launch) and there
     * is nothing left to overlap it with.  Input still on the host: batches, so that the copy of the next one runs beside
     * the kernels of this one. */
    const bool fuse = k1_fused();
    uint32_t BATCH = fuse && !h_src ? std::max<uint32_t>(nchunks, 1u) : c->batch_chunks;
    uint32_t first_env = 0;
    if (h_src) {        /* developer knobs for the host-input pipeline: chunks per batch / in the first batch */
        if (const char *e = getenv("QATZIP_AMD_HOST_BATCH")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 256) BATCH = v; }
        if (const char *e = getenv("QATZIP_AMD_HOST_FIRST")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 256) first_env = v; }
    }
    /* Host input of 64 MiB and more (level 1, chunks of the common kind): ONE launch over the whole call that starts at
     * once and takes its chunks as they land - the pieces of the copy go out behind it and the host raises the launch's
     * watermark (c->h_wm, pinned) as each completes (qzk_wait_input).  No batch boundaries, so no tails but the last one,
     * and the parse runs beside the whole copy instead of beside all but the first batch of it. */
    /* The launch fills every register file and then WAITS for the copy: that is only safe when the copy needs none of the
     * compute units and cannot stall behind the host - page-locked source memory moved by a copy engine.  A pageable
     * source (staged by the runtime piece by piece), a source that is not the caller's pinned memory, or a process that
     * has switched the copy engines off (HSA_ENABLE_SDMA=0: copies become kernels, which cannot run beside the resident
     * waves) take the batched pipeline; so does the retry of a call whose launch gave up waiting (c->no_stream_in,
     * qzd_deflate_raw_from_host). */
    bool stream_in = h_src && fuse && !cdesc && n >= (64ull << 20) && c->h_wm && !c->no_stream_in && !getenv("QATZIP_AMD_HOST_BATCHED");
    if (stream_in) {
        const char *sd = getenv("HSA_ENABLE_SDMA");
        if (sd && sd[0] == '0') stream_in = false;
    }
    if (stream_in) {
        hipPointerAttribute_t a0, a1;
        const bool p0 = hipPointerGetAttributes(&a0, h_src) == hipSuccess && a0.type == hipMemoryTypeHost;
        const bool p1 = p0 && hipPointerGetAttributes(&a1, h_src + n - 1) == hipSuccess && a1.type == hipMemoryTypeHost;
        if (!p1) { (void)hipGetLastError(); stream_in = false; }     /* (an unregistered pointer is an error to the query: cleared) */
    }
    if (stream_in) { BATCH = nchunks; first_env = 0; c->h_wm[0] = 0; c->h_wm[1] = 0; }
    /* a launch that waits for its input must hear from the host on every way out of this function: whatever returns
     * early (a HIP error between the launch and the copy) leaves the watermark at "giving up" */
    struct FedGuard {
        uint32_t *wm; bool armed;
        ~FedGuard() { if (armed && wm) __atomic_store_n(&wm[0], 0xffffffffu, __ATOMIC_RELEASE); }
    } fed_guard = { c->h_wm, stream_in };
    /* one launch for the whole call: its waves also move the stream to d_dst (qzk_outp) - no scan, no gather behind it */
    const char *oute = getenv("QATZIP_AMD_K1_OUT");
    /* (measured, profiles/r3_api_stream.txt: with the destination across PCIe the stream then travels while the parse runs,
     * 25.0 -> 27.3 GB/s for a 1 GiB qzCompress; with the destination in HBM the gather kernel's 2 ms are cheaper than the
     * waves' own copies, 115.8 vs 117.9 ms per 4 GiB - so: calls fed from the host only, QATZIP_AMD_K1_OUT=launch|gather
     * forces either) */
    const bool out_in_launch = fuse && BATCH >= nchunks && (oute ? !strcmp(oute, "launch") : stream_in);
    qzk_outp outp;
    memset(&outp, 0, sizeof(outp));
    if (out_in_launch) {
        outp.dst = d_dst; outp.cap = dst_cap; outp.offs = c->d_offs; outp.front = c->d_offs + nchunks;
        outp.pub = (uint32_t *)(c->d_offs + nchunks + 1); outp.running = c->d_running; outp.overflow = c->d_overflow;
        HIPCHK(c, hipMemsetAsync(c->d_offs, 0, QZD_OFFS_BYTES(nchunks), c->st[0]));
    }
    c->k1ev_n = 0;
    c->nbatches = (nchunks + BATCH - 1) / BATCH;

    guard.launched = true;
    HIPCHK(c, hipMemsetAsync(c->d_running, 0, 8, c->st[0]));
    HIPCHK(c, hipMemsetAsync(c->d_overflow, 0, 4, c->st[0]));
    HIPCHK(c, hipEventRecord(c->ev_begin, c->st[0]));
    HIPCHK(c, hipEventRecord(c->done[0], c->st[0]));
    HIPCHK(c, hipStreamWaitEvent(c->st[1], c->done[0], 0));

    /* with the input still on the host the first batch is one round of the persistent workgroups instead of three:
     * nothing can overlap its copy, so it is kept short */
    const uint32_t FIRST = h_src && nchunks > BATCH ? (first_env ? first_env : std::max<uint32_t>(c->k1_wgs, 1024u)) : BATCH;
    const uint32_t tab_wgs = pool->tab_wgs;
    if (FIRST != BATCH) c->nbatches = 1 + (nchunks - FIRST + BATCH - 1) / BATCH;
    for (uint32_t b = 0, k = 0, bnext = 0; b < nchunks; b = bnext, k++) {
        const int s = (int)(k % QZD_NBUF), so = (int)((k + 1) % QZD_NBUF);
        const uint32_t bsz = k == 0 ? std::min(FIRST, BATCH) : BATCH;
        const uint32_t bn = nchunks - b < bsz ? nchunks - b : bsz;
        bnext = b + bn;
        const uint64_t boff = (uint64_t)b * chunk_sz;
        const uint64_t blen = n - boff;      /* bytes from this batch's first chunk to the end of the call */
        const uint32_t final_chunk = (last && b + bn == nchunks) ? bn - 1 : ~0u;
        hipStream_t st = c->st[s];
        const bool timed = k < QZD_NBUF;     /* events of the first use of each buffer set */
        /* input still in host memory: batch k's bytes were sent while batch k-1 was enqueued (below); its kernels wait
         * for them, and batch k+1's copy goes out behind this batch's launches - asynchronous from pinned memory, and
         * from pageable memory the host blocks in it while the GPU works on batch k */
        auto send = [&](uint32_t kk, uint32_t bb) -> hipError_t {
            const uint64_t o = (uint64_t)bb * chunk_sz, len = std::min<uint64_t>((uint64_t)(kk == 0 ? bsz : BATCH) * chunk_sz, n - o);
            hipError_t e = hipMemcpyAsync((void *)(d_src + o), h_src + o, len, hipMemcpyHostToDevice, c->st_copy);
            return e != hipSuccess ? e : hipEventRecord(c->cp_ev[kk % (QZD_NBUF + 1)], c->st_copy);
        };
        if (h_src && n && !stream_in) {
            if (k == 0) HIPCHK(c, send(0, 0));
            HIPCHK(c, hipStreamWaitEvent(st, c->cp_ev[k % (QZD_NBUF + 1)], 0));
        }
        /* the K1 workgroups of consecutive batches share the per-workgroup tables, so K1 of batch k starts when K1 of
         * batch k-1 is done; what overlaps with it is K2/scan/gather of batch k-1 */
        if (k > 0) HIPCHK(c, hipStreamWaitEvent(st, c->k1done[so], 0));
        /* workgroups: one per CU at most (and per table); waves per workgroup: as many as it takes to give every chunk of
         * a round its own wave - a small launch keeps one chunk per CU */
        const uint32_t wgs = bn < tab_wgs ? bn : tab_wgs;
        const uint32_t wpw = std::min<uint32_t>((uint32_t)QZK_K1_WAVES, std::min<uint32_t>((bn + wgs - 1) / wgs, (c->k1_wgs + wgs - 1) / wgs));
        HIPCHK(c, hipMemsetAsync(c->k1_counter + s, 0, 4, st));
        if (timed) HIPCHK(c, hipEventRecord(c->ev[s][0], st));
        if (k < QZD_K1EV) HIPCHK(c, hipEventRecord(c->k1ev[k][0], st));
        /* K2 rides in the K1 waves (qzk_lz77_pull_kernel): symbols per wave, slots / meta by the chunk's number in the call */
        uint8_t *const slots_b = fuse ? pool->slots[0] + (size_t)b * stride : pool->slots[s];
        qzk_lzmeta *const meta_b = fuse ? pool->meta[0] + b : pool->meta[s];
        const int sb = fuse ? 0 : s;
        if (stream_in || out_in_launch)
            hipLaunchKernelGGL(qzk_lz77_pull_fed_kernel, dim3(wgs), dim3(64 * wpw), 0, st, d_src + boff, blen, chunk_sz, bn,
                               pool->sym_lc[sb], pool->sym_dist[sb], meta_b, pool->tables, c->k1_counter + s, cdesc ? cdesc + b : NULL,
                               pool->epoch, fuse ? slots_b : (uint8_t *)NULL, stride, final_chunk, c->d_len + b,
                               fuse ? c->d_crc + b : (uint32_t *)NULL, stream_in ? (const uint32_t *)c->h_wm : (const uint32_t *)NULL, outp);
        else
            hipLaunchKernelGGL(qzk_lz77_pull_kernel, dim3(wgs), dim3(64 * wpw), 0, st, d_src + boff, blen, chunk_sz, bn,
                               pool->sym_lc[sb], pool->sym_dist[sb], meta_b, pool->tables, c->k1_counter + s, cdesc ? cdesc + b : NULL,
                               pool->epoch, fuse ? slots_b : (uint8_t *)NULL, stride, final_chunk, c->d_len + b,
                               fuse ? c->d_crc + b : (uint32_t *)NULL, (const uint32_t *)NULL, outp);
        pool->epoch += bn;
        HIPCHK(c, hipEventRecord(c->k1done[s], st));
        if (k < QZD_K1EV) { HIPCHK(c, hipEventRecord(c->k1ev[k][1], st)); c->k1ev_chunks[k] = bn; c->k1ev_n = k + 1; }
        if (timed) HIPCHK(c, hipEventRecord(c->ev[s][1], st));
        if (!fuse)
            hipLaunchKernelGGL(qzk_huff_kernel, dim3(bn), dim3(QZK_HW), 0, st, d_src + boff, blen, chunk_sz, bn,
                               pool->sym_lc[s], pool->sym_dist[s], pool->meta[s], pool->slots[s], stride, final_chunk,
                               c->d_len + b, cdesc ? cdesc + b : NULL);
        if (!fuse)          /* fused: the chunk CRCs ride along K1's input reads */
            hipLaunchKernelGGL(qzk_crc_chunks_kernel, dim3(bn), dim3(QZK_HT), 0, st, d_src + boff, blen, chunk_sz, bn, c->d_crc + b,
                               cdesc ? cdesc + b : NULL);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[s][2], st));
        /* the running total serialises scan/gather of consecutive batches across the two streams */
        if (k > 0) HIPCHK(c, hipStreamWaitEvent(st, c->done[so], 0));
        if (!out_in_launch) {
            hipLaunchKernelGGL(qzk_scan_kernel, dim3(1), dim3(1024), 0, st, c->d_len + b, bn, c->d_offs + b, c->d_running);
            hipLaunchKernelGGL(qzk_gather_kernel, dim3(bn), dim3(256), 0, st, slots_b, stride, c->d_len + b,
                               c->d_offs + b, bn, d_dst, dst_cap, c->d_overflow);
        }
        if (timed) HIPCHK(c, hipEventRecord(c->ev[s][3], st));
        HIPCHK(c, hipEventRecord(c->done[s], st));
        if (h_src && !stream_in && bnext < nchunks) HIPCHK(c, send(k + 1, bnext));
    }
    /* join: stream 0 waits for stream 1, then publishes the totals */
    HIPCHK(c, hipStreamWaitEvent(c->st[0], c->done[1], 0));
    HIPCHK(c, hipStreamWaitEvent(c->st[0], c->done[0], 0));
    HIPCHK(c, hipMemcpyAsync(c->h_running, c->d_running, 8, hipMemcpyDeviceToHost, c->st[0]));
    HIPCHK(c, hipMemcpyAsync(c->h_overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, c->st[0]));
    HIPCHK(c, hipEventRecord(c->ev_end, c->st[0]));
    HIPCHK(c, hipGetLastError());
    if (stream_in) {
        /* the copy, in pieces (small ones first: the launch is waiting), two in flight; the watermark follows the last
         * piece known to be complete.  Whatever goes wrong, the launch is told (0xffffffff) before this returns. */
        uint64_t off = 0, done_to = 0, ends[QZD_NBUF + 1] = {0};
        hipError_t e = hipSuccess;
        uint32_t i = 0;
        const bool trace = getenv("QATZIP_AMD_STREAM_TRACE") != NULL;
        struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
        auto landed = [&](uint64_t upto) {
            __atomic_store_n(&c->h_wm[0], upto >= n ? nchunks : (uint32_t)(upto / chunk_sz), __ATOMIC_RELEASE);
            if (trace) {
                struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
                fprintf(stderr, "[stream] %8.3f ms  %6.1f MiB landed\n", (t.tv_sec - ts0.tv_sec) * 1e3 + (t.tv_nsec - ts0.tv_nsec) / 1e6, upto / 1048576.0);
            }
        };
        for (; off < n && e == hipSuccess; i++) {
            /* every wave of the launch has pulled a chunk and waits for it: fine steps while they start (the first
             * 256 MiB feed the first chunk of each of 4096 waves), coarse ones once the copy is ahead of the parse */
            const uint64_t want = i < 2 ? (4ull << 20) : off < (256ull << 20) ? (8ull << 20) : (64ull << 20);
            const uint64_t len = std::min<uint64_t>(want, n - off);
            e = hipMemcpyAsync((void *)(d_src + off), h_src + off, len, hipMemcpyHostToDevice, c->st_copy);
            if (e == hipSuccess) e = hipEventRecord(c->cp_ev[i % (QZD_NBUF + 1)], c->st_copy);
            off += len; ends[i % (QZD_NBUF + 1)] = off;
            if (i >= 1 && e == hipSuccess) {
                e = hipEventSynchronize(c->cp_ev[(i - 1) % (QZD_NBUF + 1)]);
                if (e == hipSuccess) landed(done_to = ends[(i - 1) % (QZD_NBUF + 1)]);
            }
        }
        if (e == hipSuccess && i) e = hipEventSynchronize(c->cp_ev[(i - 1) % (QZD_NBUF + 1)]);
        if (e != hipSuccess) {
            __atomic_store_n(&c->h_wm[0], 0xffffffffu, __ATOMIC_RELEASE);
            HIPCHK(c, e);
        }
        landed(n);
        (void)done_to;
    }
    fed_guard.armed = false;
    return QZD_OK;
}

#include "qzk_deflate_lazy.h"
#define QZD_LAZY_BATCH 16384u       /* chunks per round of the lazy path: 32 bytes of scratch per input byte */

/* comp_lvl 4-9: chains (L1), one wave per chunk searching 64 positions at a time (L2), serial parse over the stored
 * answers (L3), then K2 / CRC / scan / gather as everywhere else.  Rounds of QZD_LAZY_BATCH chunks on one stream. */
static int deflate_lazy_path(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level, int last,
                             uint8_t *d_dst, uint64_t dst_cap, uint32_t nchunks, const uint32_t *cdesc)
{
    const uint32_t stride = slot_stride_for(chunk_sz);
    const uint32_t B = nchunks < QZD_LAZY_BATCH ? nchunks : QZD_LAZY_BATCH;
    const size_t span = (size_t)B * chunk_sz;
    const size_t symb = (span + 511) & ~(size_t)255;
    const size_t metab = ((size_t)B * sizeof(qzk_lzmeta) + 255) & ~(size_t)255;
    const size_t slotb = ((size_t)B * stride + 255) & ~(size_t)255;
    const size_t headb = (size_t)B * QZK_HSIZE * 4, pdb = 16 * symb, resb = 8 * symb;
    const size_t need = symb * 3 + metab + slotb + headb + pdb + resb;
    if (need > c->lane_cap) {
        hipDeviceSynchronize();
        if (c->d_lane) hipFree(c->d_lane);
        c->d_lane = NULL; c->lane_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_lane, need));
        c->lane_cap = need;
    }
    if (nchunks > c->call_cap) {
        hipDeviceSynchronize();
        hipFree(c->d_len); hipFree(c->d_crc); hipFree(c->d_offs);
        c->d_len = NULL; c->d_crc = NULL; c->d_offs = NULL; c->call_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_len, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_crc, (size_t)nchunks * 4));
        HIPCHK(c, hipMalloc(&c->d_offs, QZD_OFFS_BYTES(nchunks)));
        c->call_cap = nchunks;
    }
    uint8_t *pb = c->d_lane;
    uint8_t *sym_lc = pb; pb += symb;
    uint16_t *sym_dist = (uint16_t *)pb; pb += 2 * symb;
    qzk_lzmeta *meta = (qzk_lzmeta *)pb; pb += metab;
    uint8_t *slots = pb; pb += slotb;
    uint32_t *head = (uint32_t *)pb; pb += headb;
    qzk_lazyrec *pd = (qzk_lazyrec *)pb; pb += pdb;
    qzk_lazyres *res = (qzk_lazyres *)pb;
    hipStream_t st = c->st[0];
    const qzk_lvlcfg cfg = qzk_level_cfg(level);
    c->last_nchunks = nchunks; c->nbatches = 1;
    HIPCHK(c, hipMemsetAsync(c->d_running, 0, 8, st));
    HIPCHK(c, hipMemsetAsync(c->d_overflow, 0, 4, st));
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    HIPCHK(c, hipEventRecord(c->ev[0][0], st));
    for (uint32_t b = 0; b < nchunks; b += B) {
        const uint32_t bn = nchunks - b < B ? nchunks - b : B;
        const uint64_t boff = (uint64_t)b * chunk_sz, blen = n - boff;
        const uint32_t *cd = cdesc ? cdesc + b : NULL;
        const uint32_t final_chunk = (last && b + bn == nchunks) ? bn - 1 : ~0u;
        HIPCHK(c, hipMemsetAsync(head, 0, (size_t)bn * QZK_HSIZE * 4, st));
        hipLaunchKernelGGL(qzk_lazy_chain_kernel, dim3(bn), dim3(64), 0, st, d_src + boff, blen, chunk_sz, bn, cd, head, pd);
        hipLaunchKernelGGL(qzk_lazy_search_kernel, dim3(bn), dim3(64), 0, st, d_src + boff, blen, chunk_sz, bn, cd, pd, res, cfg);
        hipLaunchKernelGGL(qzk_lazy_parse_kernel, dim3(bn), dim3(64), 0, st, d_src + boff, blen, chunk_sz, bn, cd, res,
                           sym_lc, sym_dist, meta, cfg);
        hipLaunchKernelGGL(qzk_huff_kernel, dim3(bn), dim3(QZK_HW), 0, st, d_src + boff, blen, chunk_sz, bn, sym_lc, sym_dist,
                           meta, slots, stride, final_chunk, c->d_len + b, cd);
        hipLaunchKernelGGL(qzk_crc_chunks_kernel, dim3(bn), dim3(QZK_HT), 0, st, d_src + boff, blen, chunk_sz, bn, c->d_crc + b, cd);
        hipLaunchKernelGGL(qzk_scan_kernel, dim3(1), dim3(1024), 0, st, c->d_len + b, bn, c->d_offs + b, c->d_running);
        hipLaunchKernelGGL(qzk_gather_kernel, dim3(bn), dim3(256), 0, st, slots, stride, c->d_len + b, c->d_offs + b, bn,
                           d_dst, dst_cap, c->d_overflow);
    }
    HIPCHK(c, hipEventRecord(c->ev[0][1], st));
    HIPCHK(c, hipEventRecord(c->ev[0][2], st));
    HIPCHK(c, hipEventRecord(c->ev[0][3], st));
    HIPCHK(c, hipMemcpyAsync(c->h_running, c->d_running, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(c->h_overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipGetLastError());
    return QZD_OK;
}

extern "C" int qzd_deflate_raw_async(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level,
                                     int last, uint8_t *d_dst, uint64_t dst_cap)
{
    return deflate_enqueue(c, d_src, n, chunk_sz, level, last, d_dst, dst_cap, NULL);
}

/* the same call with the input still in host memory: h_src is copied to d_stage (n bytes of device memory) batch by
 * batch, each copy behind the previous batch's launches */
extern "C" int qzd_deflate_raw_from_host(qzd_ctx *c, const uint8_t *h_src, uint8_t *d_stage, uint64_t n, uint32_t chunk_sz,
                                         int level, int last, uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len,
                                         uint32_t *h_chunk_crc)
{
    if (!c || (n && (!h_src || !d_stage))) return QZD_ERR_PARAM;
    int rc = deflate_enqueue(c, d_stage, n, chunk_sz, level, last, d_dst, dst_cap, NULL, h_src);
    if (rc) return rc;
    rc = qzd_sync(c);
    if (rc) return rc;
    /* A launch fed while it runs gives up when its input stops coming for about a second (a wave says so in h_wm[1]) or
     * when a wave waited in vain for the chunks before its own (overflow bit 1): the copy stalled - a preempted host
     * thread, a source that was paged out under it.  Nothing is lost but time: the call goes through the batched
     * pipeline, which waits for nothing. */
    const bool gave_up = (c->h_wm && c->h_wm[1]) || (*c->h_overflow & 2u);
    if (gave_up) {
        if (c->h_wm) c->h_wm[1] = 0;
        c->no_stream_in = true;
        rc = deflate_enqueue(c, d_stage, n, chunk_sz, level, last, d_dst, dst_cap, NULL, h_src);
        if (!rc) rc = qzd_sync(c);
        c->no_stream_in = false;
        if (rc) return rc;
    }
    return qzd_result(c, h_out_len, h_chunk_crc, c->last_nchunks);
}

/* Many small requests in one launch (the submission queue of qzCompress2, qz_api.cpp): d_src holds nslots slots of
 * chunk_sz bytes; a request occupies consecutive slots from a slot boundary; h_cdesc[k] = bytes in slot k, bit 31 set
 * on the slot that ends its request (it gets BFINAL, the others the flush marker).  The streams of all slots are
 * written back to back to d_dst; h_slot_len / h_slot_crc give every slot's share and CRC-32. */
extern "C" int qzd_deflate_slots(qzd_ctx *c, const uint8_t *d_src, uint32_t nslots, uint32_t chunk_sz, int level,
                                 const uint32_t *h_cdesc, uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len,
                                 uint32_t *h_slot_len, uint32_t *h_slot_crc)
{
    if (!c || !h_cdesc || !nslots) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    for (uint32_t k = 0; k < nslots; k++) if ((h_cdesc[k] & 0x7fffffffu) > chunk_sz) return QZD_ERR_PARAM;
    if (nslots > c->cdesc_cap) {
        hipDeviceSynchronize();
        if (c->d_cdesc) hipFree(c->d_cdesc);
        c->d_cdesc = NULL; c->cdesc_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_cdesc, (size_t)nslots * 4));
        c->cdesc_cap = nslots;
    }
    HIPCHK(c, hipMemcpy(c->d_cdesc, h_cdesc, (size_t)nslots * 4, hipMemcpyHostToDevice));
    int rc = deflate_enqueue(c, d_src, (uint64_t)nslots * chunk_sz, chunk_sz, level, 0, d_dst, dst_cap, c->d_cdesc);
    if (rc) return rc;
    rc = qzd_sync(c);
    if (rc) return rc;
    rc = qzd_result(c, h_out_len, h_slot_crc, nslots);
    if (rc) return rc;
    return h_slot_len ? qzd_chunk_lens(c, h_slot_len, nslots) : QZD_OK;
}

extern "C" int qzd_sync(qzd_ctx *c)
{
    if (!c) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    const hipError_t e0 = hipStreamSynchronize(c->st[0]), e1 = hipStreamSynchronize(c->st[1]), e2 = hipStreamSynchronize(c->st_copy);
    const hipError_t e3 = hipStreamSynchronize(c->st_out);
    HIPCHK(c, e0); HIPCHK(c, e1); HIPCHK(c, e2); HIPCHK(c, e3);
    for (uint32_t k = 0; k < c->k1ev_n; k++) {      /* harvest the K1 launch timings of the call that just finished */
        float t = 0;
        if (hipEventElapsedTime(&t, c->k1ev[k][0], c->k1ev[k][1]) == hipSuccess) {
            c->k1_ms_acc += t; c->k1_launch_acc++; c->k1_chunk_acc += c->k1ev_chunks[k];
        }
    }
    c->k1ev_n = 0;
    return QZD_OK;
}

extern "C" int qzd_k1_stats(qzd_ctx *c, double *ms, uint64_t *launches, uint64_t *chunks, int reset)
{
    if (!c) return QZD_ERR_PARAM;
    if (ms) *ms = c->k1_ms_acc;
    if (launches) *launches = c->k1_launch_acc;
    if (chunks) *chunks = c->k1_chunk_acc;
    if (reset) { c->k1_ms_acc = 0; c->k1_launch_acc = 0; c->k1_chunk_acc = 0; }
    return QZD_OK;
}

extern "C" int qzd_result(qzd_ctx *c, uint64_t *h_out_len, uint32_t *h_chunk_crc, uint32_t nchunks)
{
    if (!c) return QZD_ERR_PARAM;
    if (*c->h_overflow & 2u) { snprintf(c->err, sizeof(c->err), "a wave waited in vain for the chunks before its own"); return QZD_ERR_HIP; }
    if (*c->h_overflow) { snprintf(c->err, sizeof(c->err), "destination too small"); return QZD_ERR_DSTCAP; }
    if (h_out_len) *h_out_len = *c->h_running;
    if (h_chunk_crc) {
        if (nchunks > c->last_nchunks) nchunks = c->last_nchunks;
        HIPCHK(c, hipMemcpy(h_chunk_crc, c->d_crc, (size_t)nchunks * 4, hipMemcpyDeviceToHost));
    }
    return QZD_OK;
}

extern "C" int qzd_deflate_raw(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level, int last,
                               uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_chunk_crc)
{
    int rc = qzd_deflate_raw_async(c, d_src, n, chunk_sz, level, last, d_dst, dst_cap);
    if (rc) return rc;
    rc = qzd_sync(c);
    if (rc) return rc;
    return qzd_result(c, h_out_len, h_chunk_crc, c->last_nchunks);
}

/* compressed size of every chunk of the last deflate call (for partial-progress reporting) */
extern "C" int qzd_chunk_lens(qzd_ctx *c, uint32_t *h_len, uint32_t nchunks)
{
    if (!c || !h_len) return QZD_ERR_PARAM;
    hipSetDevice(c->device);
    if (nchunks > c->last_nchunks) nchunks = c->last_nchunks;
    HIPCHK(c, hipMemcpy(h_len, c->d_len, (size_t)nchunks * 4, hipMemcpyDeviceToHost));
    return QZD_OK;
}

extern "C" int qzd_last_timing(qzd_ctx *c, float ms[4])
{
    if (!c || !ms) return QZD_ERR_PARAM;
    float t = 0;
    ms[0] = ms[1] = ms[2] = ms[3] = 0;
    uint32_t nb = c->nbatches < QZD_NBUF ? c->nbatches : QZD_NBUF;
    for (uint32_t s = 0; s < nb; s++)
        for (int k = 0; k < 3; k++) {
            if (hipEventElapsedTime(&t, c->ev[s][k], c->ev[s][k + 1]) == hipSuccess) ms[k] += t;
        }
    if (hipEventElapsedTime(&t, c->ev_begin, c->ev_end) == hipSuccess) ms[3] = t;
    return QZD_OK;
}

/* ------------------------------------------------------------------ LZ4 frames (K4 / K5) */
#include "qzk_lz4.h"

/* every frame_sz bytes of d_src become one LZ4 frame (what one qzCompress call of an LZ4 session emits for
 * src_len <= 64 KB, src/qatzip_sw.c:443-471); frames are written back to back to d_dst */
static int lz4_compress_frames_impl(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t frame_sz,
                                    uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_frame_len, uint32_t hw_hdr);
extern "C" int qzd_lz4_compress_frames(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t frame_sz,
                                       uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_frame_len)
{ return lz4_compress_frames_impl(c, d_src, n, frame_sz, d_dst, dst_cap, h_out_len, h_frame_len, 0); }
/* the same frames behind the header of the reference's hardware path (FLG 0x4C, content size = the chunk's bytes,
 * src/qatzip_lz4.c:104-132): what a QAT box writes per hw_buff_sz chunk of an LZ4 session */
extern "C" int qzd_lz4_compress_frames_hw(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t frame_sz,
                                          uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_frame_len)
{ return lz4_compress_frames_impl(c, d_src, n, frame_sz, d_dst, dst_cap, h_out_len, h_frame_len, 1); }
static int lz4_compress_frames_impl(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint32_t frame_sz,
                                    uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_frame_len, uint32_t hw_hdr)
{
    if (!c || !d_dst || (n && !d_src) || !h_out_len) return QZD_ERR_PARAM;
    /* above 64 KB a frame has several linked blocks: only the hardware path's per-chunk frames are made that way here (one
     * call = one frame of the software path goes through qzd_lz4_compress_linked) */
    const bool linked = frame_sz > QZK_LZ4_MAXBLK;
    if (frame_sz == 0 || (linked && (!hw_hdr || frame_sz > (1u << 30)))) { snprintf(c->err, sizeof(c->err), "LZ4 frames above 64 KB (linked blocks) are not produced"); return QZD_ERR_UNSUPPORTED; }
    hipSetDevice(c->device);
    const uint32_t nfr = n ? (uint32_t)((n + frame_sz - 1) / frame_sz) : 1;
    const uint32_t stride = (frame_sz + 15 + 4 * ((frame_sz + 65535) >> 16) + 8 + 64 + 15) & ~15u;
    if (nfr > c->call_cap) {
        hipDeviceSynchronize();
        hipFree(c->d_len); hipFree(c->d_crc); hipFree(c->d_offs);
        c->d_len = NULL; c->d_crc = NULL; c->d_offs = NULL; c->call_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_len, (size_t)nfr * 4));
        HIPCHK(c, hipMalloc(&c->d_crc, (size_t)nfr * 4));
        HIPCHK(c, hipMalloc(&c->d_offs, (size_t)nfr * 8));
        c->call_cap = nfr;
    }
    uint32_t batch = nfr < 16384u ? nfr : 16384u;
    if (linked && (uint64_t)batch * stride > (1ull << 30)) batch = std::max<uint32_t>(1u, (uint32_t)((1ull << 30) / stride));   /* slots of a gigabyte at most */
    if ((size_t)batch * stride > c->slot_cap) {
        hipDeviceSynchronize();
        for (int i = 0; i < QZD_NBUF; i++) { hipFree(c->slots[i]); c->slots[i] = NULL; }
        c->slot_cap = 0;
        for (int i = 0; i < QZD_NBUF; i++) HIPCHK(c, hipMalloc(&c->slots[i], (size_t)batch * stride));
        c->slot_cap = (size_t)batch * stride;
    }
    hipStream_t st = c->st[0];
    HIPCHK(c, hipMemsetAsync(c->d_running, 0, 8, st));
    HIPCHK(c, hipMemsetAsync(c->d_overflow, 0, 4, st));
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    for (uint32_t b = 0; b < nfr; b += batch) {
        const uint32_t bn = nfr - b < batch ? nfr - b : batch;
        const uint64_t boff = (uint64_t)b * frame_sz;
        /* many frames: persistent waves with their hash tables in device memory, as many per CU as it has wave slots
         * (QATZIP_AMD_LZ4_WPC=<waves per CU>, 0 = the one-launch-per-frame kernel with the table in LDS) */
        uint32_t wpc = 32;          /* measured 24 / 32: 19.6 / 21.6 GB/s (table in LDS, 8 waves: 13.2) - profiles/r3_lz4_ring_experiment.txt */
        if (const char *e = getenv("QATZIP_AMD_LZ4_WPC")) wpc = (uint32_t)atoi(e);
        if (wpc > 32) wpc = 32;
        const uint32_t cus = c->cus ? c->cus : 256u;
        if (linked) {
            /* one wave per chunk; the call's last chunk is a one-block frame when it is 64 KB or less */
            const bool tail_small = b + bn == nfr && n - (uint64_t)(nfr - 1) * frame_sz <= QZK_LZ4_MAXBLK;
            const uint32_t nl = tail_small ? bn - 1 : bn;
            if (nl) hipLaunchKernelGGL(qzk_lz4c_linked_many_kernel, dim3(nl), dim3(64), 0, st, d_src + boff, n - boff, frame_sz, nl, c->slots[0], stride, c->d_len + b);
            if (tail_small) {
                const uint64_t toff = (uint64_t)(nfr - 1) * frame_sz;
                hipLaunchKernelGGL(qzk_lz4c_kernel, dim3(1), dim3(64), 0, st, d_src + toff, n - toff, (uint32_t)QZK_LZ4_MAXBLK, 1u, c->slots[0] + (size_t)(bn - 1) * stride, stride, c->d_len + nfr - 1, hw_hdr);
            }
        } else if (wpc && bn > 8 * cus) {
            const uint32_t waves = std::min<uint32_t>(bn, wpc * cus);
            if (waves > c->lz4tab_waves) {
                hipDeviceSynchronize();
                if (c->d_lz4tab) hipFree(c->d_lz4tab);
                c->d_lz4tab = NULL; c->lz4tab_waves = 0;
                /* frame counter (256 B), the waves' epochs, their tables (8192 entries of 8 bytes each: 64 KiB a wave, 512 MiB for
                 * a full device) - cleared ONCE here: the epochs take the place of a clearing per frame */
                const size_t epb = ((size_t)waves * 4 + 255) & ~(size_t)255, tb = (size_t)waves * QZK_L4C_TABW * 8;
                HIPCHK(c, hipMalloc(&c->d_lz4tab, 256 + epb + tb));
                HIPCHK(c, hipMemsetAsync(c->d_lz4tab, 0, 256 + epb + tb, st));
                c->lz4tab_waves = waves;
            }
            HIPCHK(c, hipMemsetAsync(c->d_lz4tab, 0, 4, st));
            {
                const size_t epb = ((size_t)c->lz4tab_waves * 4 + 255) & ~(size_t)255;
                hipLaunchKernelGGL(qzk_lz4c_pull_kernel, dim3(waves), dim3(64), 0, st, d_src + boff, n - boff, frame_sz, bn, c->slots[0], stride,
                                   c->d_len + b, hw_hdr, (uint64_t *)(c->d_lz4tab + 256 + epb), (uint32_t *)(c->d_lz4tab + 256), (uint32_t *)c->d_lz4tab);
            }
        } else
            hipLaunchKernelGGL(qzk_lz4c_kernel, dim3(bn), dim3(64), 0, st, d_src + boff, n - boff, frame_sz, bn, c->slots[0], stride, c->d_len + b, hw_hdr);
        hipLaunchKernelGGL(qzk_scan_kernel, dim3(1), dim3(1024), 0, st, c->d_len + b, bn, c->d_offs + b, c->d_running);
        hipLaunchKernelGGL(qzk_gather_kernel, dim3(bn), dim3(256), 0, st, c->slots[0], stride, c->d_len + b, c->d_offs + b, bn, d_dst, dst_cap, c->d_overflow);
    }
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipMemcpyAsync(c->h_running, c->d_running, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(c->h_overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    c->last_nchunks = nfr;
    if (*c->h_overflow) { snprintf(c->err, sizeof(c->err), "destination too small"); return QZD_ERR_DSTCAP; }
    *h_out_len = *c->h_running;
    if (h_frame_len) HIPCHK(c, hipMemcpy(h_frame_len, c->d_len, (size_t)nfr * 4, hipMemcpyDeviceToHost));
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev_begin, c->ev_end) == hipSuccess) c->ms[3] = t;
    return QZD_OK;
}

/* one call above 64 KB as the ONE frame with linked blocks that LZ4F_compressFrame writes for it (one wave, serial) */
extern "C" int qzd_lz4_compress_linked(qzd_ctx *c, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                                       uint64_t *h_out_len)
{
    if (!c || !d_src || !d_dst || !h_out_len) return QZD_ERR_PARAM;
    if (n <= QZK_LZ4_MAXBLK || n > 0x7fff0000ull) { snprintf(c->err, sizeof(c->err), "linked LZ4 frames: 64 KB < n <= 0x7fff0000"); return QZD_ERR_UNSUPPORTED; }
    const uint64_t bound = 19 + 4 * ((n + 65535) >> 16) + n + 8;
    if (dst_cap < bound) { snprintf(c->err, sizeof(c->err), "destination too small"); return QZD_ERR_DSTCAP; }
    hipSetDevice(c->device);
    hipStream_t st = c->st[0];
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    hipLaunchKernelGGL(qzk_lz4c_linked_kernel, dim3(1), dim3(64), 0, st, d_src, (uint32_t)n, d_dst, (uint32_t *)c->d_overflow);
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipMemcpyAsync(c->h_overflow, c->d_overflow, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    *h_out_len = *c->h_overflow;
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev_begin, c->ev_end) == hipSuccess) c->ms[3] = t;
    return QZD_OK;
}

/* decode nsegs LZ4 frames {u64 in_off, u64 out_off, u32 in_len, u32 out_cap} -> {i32 status, u32 in_used, u32 out_len, u32 pad};
 * content checksums are verified on the GPU (XXH32) */
extern "C" int qzd_lz4_decompress_frames(qzd_ctx *c, const uint8_t *d_comp, uint8_t *d_out, const void *h_segs,
                                         uint32_t nsegs, void *h_res)
{
    if (!c || !h_segs || !h_res) return QZD_ERR_PARAM;
    if (nsegs == 0) return QZD_OK;
    hipSetDevice(c->device);
    const size_t sb = (size_t)nsegs * sizeof(qzk_lz4seg), rb = (size_t)nsegs * sizeof(qzk_lz4res);
    int rc = qzd_aux_reserve(c, sb + rb + 64);
    if (rc) return rc;
    qzk_lz4seg *d_segs = (qzk_lz4seg *)c->d_aux;
    qzk_lz4res *d_res = (qzk_lz4res *)(c->d_aux + ((sb + 15) & ~(size_t)15));
    hipStream_t st = c->st[0];
    HIPCHK(c, hipMemcpyAsync(d_segs, h_segs, sb, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_begin, st));
    /* single-wave workgroups: a workgroup leaves with its slowest frame (the resolve kernel of the deflate side: -12 %) */
    hipLaunchKernelGGL(qzk_lz4d_kernel, dim3(nsegs), dim3(64), 0, st, d_comp, d_out, d_segs, d_res, nsegs);
    HIPCHK(c, hipEventRecord(c->ev_end, st));
    HIPCHK(c, hipMemcpyAsync(h_res, d_res, rb, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev_begin, c->ev_end) == hipSuccess) c->ms[3] = t;
    return QZD_OK;
}

/* developer aid (QATZIP_AMD_WIDE_PROF=1): the phase clocks the workgroup-per-chunk parse left for the chunks of the last
 * call, 16 words per chunk (qzk_deflate_wide.h); returns the number of chunks copied */
extern "C" int qzd_debug_wide_prof(qzd_ctx *c, uint64_t *h_out, uint32_t nchunks)
{
    if (!c || !h_out || !g_wide_prof) return 0;
    hipSetDevice(c->device);
    if (nchunks > g_wide_prof_n) nchunks = g_wide_prof_n;
    if (hipMemcpy(h_out, g_wide_prof, (size_t)nchunks * 128, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return (int)nchunks;
}

#ifdef QZK_PROF
/* profiling builds only: raw K1 metadata (with per-phase cycle counters) of buffer set `s` */
extern "C" int qzd_debug_meta(qzd_ctx *c, int s, void *h_out, uint32_t nchunks)
{
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpy(h_out, g_k1pool[c->device % QZD_MAX_DEVICES].meta[s], (size_t)nchunks * sizeof(qzk_lzmeta), hipMemcpyDeviceToHost));
    return (int)sizeof(qzk_lzmeta);
}
#endif
