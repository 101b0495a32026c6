/*
 * qzd_shard.hip — ONE logical gzip-ext member built by several GPUs (one process per GPU).
 *
 * The reference's engine retires the chunks of a call in order and folds their checksums into the member's trailer
 * (doCompressOut, src/qatzip.c:1691-1718: payload memcpy + crc32_combine + footer).  Chunks are independent, so a
 * buffer shards over GPUs by contiguous chunk ranges; what is left of that retire step across GPUs is
 *   - a 32-byte record per rank (raw bytes, compressed bytes, CRC-32 of the shard), from which every rank derives the
 *     offset of its compressed shard (exclusive scan) and the root folds the trailer,
 *   - the variable-size gather of the compressed shards into the root's HBM.
 * Both travel as peer-to-peer copies into a WINDOW in the root GPU's memory that every rank has mapped through HIP IPC
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle): between GPUs of one node those are xGMI writes, straight from the
 * producer's HBM to the root's, no host bounce and no collective library - the exchange is a flat gather.
 * The only thing the ranks need from their launcher is to pass the root's 64-byte handle around once.
 *
 * Window layout:  [ world x 32-byte records | 24 bytes for the gzip-ext header | payload ... | 8 bytes trailer ]
 * A rank's record carries the sequence number of the stream it belongs to; a rank waits for the records of the ranks
 * before it (its offset), copies its shard, then raises its `done` word; the root waits for every `done`.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <new>

#include "qzd_internal.h"

typedef struct { uint64_t raw_len, comp_len; uint32_t crc, seq, done, pad; } qzd_shard_rec;    /* 32 bytes */

struct qzd_shard {
    qzd_ctx *ctx;
    uint32_t rank, world;
    uint64_t cap;                   /* payload capacity of the window */
    uint8_t *win;                   /* the window, in this process's address space (root: its own allocation) */
    bool owner;
};

#define QZD_SHARD_HDR 24u
static size_t win_bytes(uint32_t world, uint64_t cap) { return (size_t)world * sizeof(qzd_shard_rec) + QZD_SHARD_HDR + cap + 8 + 256; }
static uint8_t *win_payload(qzd_shard *s) { return s->win + (size_t)s->world * sizeof(qzd_shard_rec) + QZD_SHARD_HDR; }

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

extern "C" int qzd_shard_root_create(qzd_ctx *c, uint32_t world, uint64_t cap_bytes, uint8_t handle_out[64], qzd_shard **out)
{
    if (!c || !out || !handle_out || world == 0 || world > 4096) return QZD_ERR_PARAM;
    *out = NULL;
    hipSetDevice(c->device);
    qzd_shard *s = new (std::nothrow) qzd_shard();
    if (!s) return QZD_ERR_HIP;
    s->ctx = c; s->rank = 0; s->world = world; s->cap = cap_bytes; s->owner = true; s->win = NULL;
    const size_t nb = win_bytes(world, cap_bytes);
    if (hipMalloc(&s->win, nb) != hipSuccess) { delete s; return QZD_ERR_HIP; }
    hipMemset(s->win, 0, (size_t)world * sizeof(qzd_shard_rec) + QZD_SHARD_HDR);
    hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the 64 bytes of the ABI");
    hipError_t e = hipIpcGetMemHandle(&h, s->win);
    if (e != hipSuccess) {
        snprintf(c->err, sizeof(c->err), "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
        hipFree(s->win); delete s; return QZD_ERR_HIP;
    }
    memset(handle_out, 0, 64);
    memcpy(handle_out, &h, sizeof(h));
    *out = s;
    return QZD_OK;
}

extern "C" int qzd_shard_attach(qzd_ctx *c, uint32_t rank, uint32_t world, const uint8_t handle[64], uint64_t cap_bytes, qzd_shard **out)
{
    if (!c || !out || !handle || rank == 0 || rank >= world) return QZD_ERR_PARAM;
    *out = NULL;
    hipSetDevice(c->device);
    qzd_shard *s = new (std::nothrow) qzd_shard();
    if (!s) return QZD_ERR_HIP;
    s->ctx = c; s->rank = rank; s->world = world; s->cap = cap_bytes; s->owner = false; s->win = NULL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle((void **)&s->win, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
        snprintf(c->err, sizeof(c->err), "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
        delete s; return QZD_ERR_HIP;
    }
    *out = s;
    return QZD_OK;
}

extern "C" void qzd_shard_close(qzd_shard *s)
{
    if (!s) return;
    hipSetDevice(s->ctx->device);
    hipDeviceSynchronize();
    if (s->owner) hipFree(s->win); else hipIpcCloseMemHandle(s->win);
    delete s;
}

/* every rank, the root included: publish my record, find my offset, send my shard.  seq (!= 0) names the stream: the
 * window is reused for the next one with the next number.  timeout_s bounds every wait. */
extern "C" int qzd_shard_put(qzd_shard *s, const uint8_t *d_comp, uint64_t comp_len, uint64_t raw_len, uint32_t crc32,
                             uint32_t seq, double timeout_s, uint64_t *h_offset)
{
    if (!s || seq == 0 || (comp_len && !d_comp)) return QZD_ERR_PARAM;
    qzd_ctx *c = s->ctx;
    hipSetDevice(c->device);
    qzd_shard_rec *recs = (qzd_shard_rec *)s->win;
    qzd_shard_rec mine; mine.raw_len = raw_len; mine.comp_len = comp_len; mine.crc = crc32; mine.seq = seq; mine.done = 0; mine.pad = 0;
    HIPCHK(c, hipMemcpy(recs + s->rank, &mine, sizeof(mine), hipMemcpyHostToDevice));
    /* exclusive scan over the ranks before me: poll their records until they carry this stream's number */
    uint64_t off = 0;
    if (s->rank) {
        qzd_shard_rec *h = (qzd_shard_rec *)malloc((size_t)s->rank * sizeof(qzd_shard_rec));
        if (!h) return QZD_ERR_HIP;
        const double t0 = now_s();
        for (;;) {
            hipError_t e = hipMemcpy(h, recs, (size_t)s->rank * sizeof(qzd_shard_rec), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free(h); snprintf(c->err, sizeof(c->err), "shard poll: %s", hipGetErrorString(e)); return QZD_ERR_HIP; }
            bool all = true;
            for (uint32_t r = 0; r < s->rank; r++) if (h[r].seq != seq) all = false;
            if (all) break;
            if (now_s() - t0 > timeout_s) { free(h); snprintf(c->err, sizeof(c->err), "shard %u: the ranks before it never published stream %u", s->rank, seq); return QZD_ERR_HIP; }
            struct timespec ts = {0, 20000}; nanosleep(&ts, NULL);
        }
        for (uint32_t r = 0; r < s->rank; r++) off += h[r].comp_len;
        free(h);
    }
    if (off + comp_len > s->cap) { snprintf(c->err, sizeof(c->err), "shard window too small"); return QZD_ERR_DSTCAP; }
    /* the gather: my compressed shard goes where it belongs in the root's HBM (a peer write over xGMI) */
    if (comp_len) HIPCHK(c, hipMemcpyAsync(win_payload(s) + off, d_comp, comp_len, hipMemcpyDeviceToDevice, c->st[0]));
    HIPCHK(c, hipStreamSynchronize(c->st[0]));
    const uint32_t done = seq;
    HIPCHK(c, hipMemcpy((uint8_t *)(recs + s->rank) + offsetof(qzd_shard_rec, done), &done, 4, hipMemcpyHostToDevice));
    if (h_offset) *h_offset = off;
    return QZD_OK;
}

/* root only: wait until every shard has arrived, fold the trailer (crc32_combine in rank order, ISIZE mod 2^32), write
 * the gzip-ext header with both sizes in front of the payload and the trailer behind it.  *d_stream points at the
 * finished member inside the window (valid until the next stream or qzd_shard_close). */
extern "C" int qzd_shard_finish(qzd_shard *s, uint32_t seq, double timeout_s, uint8_t **d_stream, uint64_t *stream_len,
                                uint32_t *crc_out, uint64_t *raw_total)
{
    if (!s || s->rank != 0 || !d_stream || !stream_len) return QZD_ERR_PARAM;
    qzd_ctx *c = s->ctx;
    hipSetDevice(c->device);
    qzd_shard_rec *recs = (qzd_shard_rec *)s->win;
    qzd_shard_rec *h = (qzd_shard_rec *)malloc((size_t)s->world * sizeof(qzd_shard_rec));
    if (!h) return QZD_ERR_HIP;
    const double t0 = now_s();
    for (;;) {
        hipError_t e = hipMemcpy(h, recs, (size_t)s->world * sizeof(qzd_shard_rec), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(h); snprintf(c->err, sizeof(c->err), "shard poll: %s", hipGetErrorString(e)); return QZD_ERR_HIP; }
        bool all = true;
        for (uint32_t r = 0; r < s->world; r++) if (h[r].seq != seq || h[r].done != seq) all = false;
        if (all) break;
        if (now_s() - t0 > timeout_s) { free(h); snprintf(c->err, sizeof(c->err), "shard root: stream %u incomplete after %.0f s", seq, timeout_s); return QZD_ERR_HIP; }
        struct timespec ts = {0, 20000}; nanosleep(&ts, NULL);
    }
    uint64_t raw = 0, comp = 0; uint32_t crc = 0;
    for (uint32_t r = 0; r < s->world; r++) {
        crc = r == 0 ? h[r].crc : qzd_crc32_combine(crc, h[r].crc, h[r].raw_len);
        raw += h[r].raw_len; comp += h[r].comp_len;
    }
    free(h);
    if (raw > 0xffffffffull || comp > 0xffffffffull) { snprintf(c->err, sizeof(c->err), "a gzip-ext member holds less than 4 GiB"); return QZD_ERR_PARAM; }
    /* the header of the software path's GZIP_EXT member (src/qatzip_sw.c:61-75,158-166) with both sizes, and its trailer */
    unsigned char hdr[24] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 4, 255, 12, 0, 'Q', 'Z', 8, 0}, tr[8];
    for (int i = 0; i < 4; i++) { hdr[16 + i] = (unsigned char)(raw >> (8 * i)); hdr[20 + i] = (unsigned char)(comp >> (8 * i)); }
    for (int i = 0; i < 4; i++) { tr[i] = (unsigned char)(crc >> (8 * i)); tr[4 + i] = (unsigned char)(raw >> (8 * i)); }
    uint8_t *pay = win_payload(s);
    HIPCHK(c, hipMemcpy(pay - QZD_SHARD_HDR, hdr, 24, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(pay + comp, tr, 8, hipMemcpyHostToDevice));
    *d_stream = pay - QZD_SHARD_HDR; *stream_len = QZD_SHARD_HDR + comp + 8;
    if (crc_out) *crc_out = crc;
    if (raw_total) *raw_total = raw;
    return QZD_OK;
}
