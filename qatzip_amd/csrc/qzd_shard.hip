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

/* Round 5: the window is several allocations, none of them large.  The eight-rank rehearsal (bench.py --gpus 8 on one GPU)
 * stood still in hipIpcOpenMemHandle as soon as the one window of world x shard bytes passed 2 GiB (2049 MiB: every rank
 * but the root never came back; 512 MiB and 1.1 GiB windows had worked) - and BASELINE config 5's window would have been
 * 8 x 575 MB.  Now: a small CONTROL window (the records) that every rank maps, one SLOT per non-root rank (a shard's
 * worth, < 600 MB for the largest member a gzip-ext header can describe) that only its rank maps, and the member itself
 * in plain device memory of the root, put together by device-to-device copies out of the slots when all have arrived
 * (2 ms for 4.6 GB at the HBM's rate). */
struct qzd_shard {
    qzd_ctx *ctx;
    uint32_t rank, world;
    uint64_t cap;                   /* payload capacity of a member */
    uint64_t slot_cap;              /* ... and of one rank's slot */
    uint8_t *win;                   /* the control window, in this process's address space (root: its own allocation) */
    uint8_t *member;                /* root: [24-byte header | payload cap | 8-byte trailer] */
    uint8_t **slots;                /* root: world pointers (slot 0 unused: the root's shard goes straight into the member) */
    uint8_t *myslot;                /* a non-root rank's own slot, mapped */
    bool owner;
};

#define QZD_SHARD_HDR 24u
/* two sets of records, used alternately by the streams (seq & 1): ranks run one stream ahead of each other at most - a
 * rank publishes its record for stream n + 1 as soon as its own shard is coded, while the root may still be collecting
 * stream n, but nobody gets past waiting for the root's record of n + 1, which the root writes when n is closed (its
 * slots read out) - so a slot is never written while the root still reads it */
#define QZD_SHARD_SETS 2u
static size_t ctl_bytes(uint32_t world) { return (size_t)QZD_SHARD_SETS * world * sizeof(qzd_shard_rec) + 256; }
static qzd_shard_rec *win_recs(qzd_shard *s, uint32_t seq) { return (qzd_shard_rec *)s->win + (size_t)(seq & (QZD_SHARD_SETS - 1)) * s->world; }

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void shard_free(qzd_shard *s)
{
    if (!s) return;
    if (s->owner) {
        if (s->win) hipFree(s->win);
        if (s->member) hipFree(s->member);
        if (s->slots) { for (uint32_t r = 1; r < s->world; r++) if (s->slots[r]) hipFree(s->slots[r]); free(s->slots); }
    } else {
        if (s->win) hipIpcCloseMemHandle(s->win);
        if (s->myslot) hipIpcCloseMemHandle(s->myslot);
    }
    delete s;
}

static int ipc_handle(qzd_ctx *c, void *p, uint8_t out[64])
{
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the 64 bytes of the ABI");
    hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        snprintf(c->err, sizeof(c->err), "hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
        return QZD_ERR_HIP;
    }
    memset(out, 0, 64);
    memcpy(out, &h, sizeof(h));
    return QZD_OK;
}

extern "C" int qzd_shard_root_create(qzd_ctx *c, uint32_t world, uint64_t cap_bytes, uint8_t handle_out[64], qzd_shard **out)
{
    if (!c || !out || !handle_out || world == 0 || world > 4096) return QZD_ERR_PARAM;
    *out = NULL;
    hipSetDevice(c->device);
    qzd_shard *s = new (std::nothrow) qzd_shard();
    if (!s) return QZD_ERR_HIP;
    s->ctx = c; s->rank = 0; s->world = world; s->cap = cap_bytes; s->slot_cap = (cap_bytes + world - 1) / world;
    s->owner = true; s->win = NULL; s->member = NULL; s->myslot = NULL;
    s->slots = (uint8_t **)calloc(world, sizeof(uint8_t *));
    bool ok = s->slots != NULL && hipMalloc(&s->win, ctl_bytes(world)) == hipSuccess &&
              hipMalloc(&s->member, QZD_SHARD_HDR + cap_bytes + 8 + 256) == hipSuccess;
    for (uint32_t r = 1; ok && r < world; r++) ok = hipMalloc(&s->slots[r], s->slot_cap + 256) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); snprintf(c->err, sizeof(c->err), "shard window: device memory for %u slots of %llu bytes", world, (unsigned long long)s->slot_cap); shard_free(s); return QZD_ERR_HIP; }
    hipMemset(s->win, 0, ctl_bytes(world));
    hipDeviceSynchronize();
    if (ipc_handle(c, s->win, handle_out) != QZD_OK) { shard_free(s); return QZD_ERR_HIP; }
    *out = s;
    return QZD_OK;
}

/* root: the handle of rank r's slot (1 <= r < world), for that rank's qzd_shard_attach_slot() */
extern "C" int qzd_shard_slot_handle(qzd_shard *s, uint32_t rank, uint8_t handle_out[64])
{
    if (!s || !s->owner || !handle_out || rank == 0 || rank >= s->world) return QZD_ERR_PARAM;
    hipSetDevice(s->ctx->device);
    return ipc_handle(s->ctx, s->slots[rank], handle_out);
}

extern "C" int qzd_shard_attach(qzd_ctx *c, uint32_t rank, uint32_t world, const uint8_t handle[64], uint64_t cap_bytes, qzd_shard **out)
{
    if (!c || !out || !handle || rank == 0 || rank >= world) return QZD_ERR_PARAM;
    *out = NULL;
    hipSetDevice(c->device);
    qzd_shard *s = new (std::nothrow) qzd_shard();
    if (!s) return QZD_ERR_HIP;
    s->ctx = c; s->rank = rank; s->world = world; s->cap = cap_bytes; s->slot_cap = (cap_bytes + world - 1) / world;
    s->owner = false; s->win = NULL; s->member = NULL; s->slots = NULL; s->myslot = NULL;
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

/* a non-root rank: map its own slot (the handle the root made with qzd_shard_slot_handle for this rank) */
extern "C" int qzd_shard_attach_slot(qzd_shard *s, const uint8_t handle[64])
{
    if (!s || s->owner || !handle || s->myslot) return QZD_ERR_PARAM;
    hipSetDevice(s->ctx->device);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle((void **)&s->myslot, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { s->myslot = NULL; snprintf(s->ctx->err, sizeof(s->ctx->err), "hipIpcOpenMemHandle (slot): %s", hipGetErrorString(e)); return QZD_ERR_HIP; }
    return QZD_OK;
}

extern "C" void qzd_shard_close(qzd_shard *s)
{
    if (!s) return;
    hipSetDevice(s->ctx->device);
    hipDeviceSynchronize();
    shard_free(s);
}

/* every rank, the root included: publish my record, wait for the ranks before me (my offset; and the root's record is what
 * says that the stream before this one has been read out of the slots), send my shard.  seq (!= 0) names the stream: the
 * window is reused for the next one with the next number.  timeout_s bounds every wait. */
extern "C" int qzd_shard_put(qzd_shard *s, const uint8_t *d_comp, uint64_t comp_len, uint64_t raw_len, uint32_t crc32,
                             uint32_t seq, double timeout_s, uint64_t *h_offset)
{
    if (!s || seq == 0 || (comp_len && !d_comp)) return QZD_ERR_PARAM;
    qzd_ctx *c = s->ctx;
    hipSetDevice(c->device);
    if (s->rank && !s->myslot) { snprintf(c->err, sizeof(c->err), "shard %u: its slot was never attached", s->rank); return QZD_ERR_PARAM; }
    qzd_shard_rec *recs = win_recs(s, seq);
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
    if (off + comp_len > s->cap || (s->rank && comp_len > s->slot_cap)) { snprintf(c->err, sizeof(c->err), "shard window too small"); return QZD_ERR_DSTCAP; }
    /* the gather: my compressed shard goes to the root's HBM - into my slot (a peer write over xGMI); the root's own
     * straight to the head of the member */
    if (comp_len) HIPCHK(c, hipMemcpyAsync(s->rank ? s->myslot : s->member + QZD_SHARD_HDR, d_comp, comp_len, hipMemcpyDeviceToDevice, c->st[0]));
    HIPCHK(c, hipStreamSynchronize(c->st[0]));
    const uint32_t done = seq;
    HIPCHK(c, hipMemcpy((uint8_t *)(recs + s->rank) + offsetof(qzd_shard_rec, done), &done, 4, hipMemcpyHostToDevice));
    if (h_offset) *h_offset = off;
    return QZD_OK;
}

/* root only: wait until every shard has arrived, put them together (slot r to its offset in the member), fold the trailer
 * (crc32_combine in rank order, ISIZE mod 2^32), write the gzip-ext header with both sizes in front of the payload and the
 * trailer behind it.  *d_stream points at the finished member (valid until the next stream or qzd_shard_close). */
/* the member's 24-byte gzip-ext header with both sizes (src/qatzip_sw.c:61-75,158-166; XFL follows the level as zlib's own
 * gzip header does: 4 = fastest, 2 = best, 0 otherwise) and its trailer */
static void member_frame(unsigned char hdr[24], unsigned char tr[8], int level, uint64_t raw, uint64_t comp, uint32_t crc)
{
    static const unsigned char h0[24] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 4, 255, 12, 0, 'Q', 'Z', 8, 0};
    memcpy(hdr, h0, 24);
    hdr[8] = level == 9 ? 2 : level < 2 ? 4 : 0;
    for (int i = 0; i < 4; i++) { hdr[16 + i] = (unsigned char)(raw >> (8 * i)); hdr[20 + i] = (unsigned char)(comp >> (8 * i)); }
    for (int i = 0; i < 4; i++) { tr[i] = (unsigned char)(crc >> (8 * i)); tr[4 + i] = (unsigned char)(raw >> (8 * i)); }
}

extern "C" int qzd_shard_finish(qzd_shard *s, uint32_t seq, double timeout_s, int level, uint8_t **d_stream, uint64_t *stream_len,
                                uint32_t *crc_out, uint64_t *raw_total)
{
    if (!s || s->rank != 0 || !d_stream || !stream_len) return QZD_ERR_PARAM;
    qzd_ctx *c = s->ctx;
    hipSetDevice(c->device);
    qzd_shard_rec *recs = win_recs(s, seq);
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
    uint8_t *pay = s->member + QZD_SHARD_HDR;
    for (uint32_t r = 0; r < s->world; r++) {
        crc = r == 0 ? h[r].crc : qzd_crc32_combine(crc, h[r].crc, h[r].raw_len);
        if (r && h[r].comp_len) {
            if (comp + h[r].comp_len > s->cap || h[r].comp_len > s->slot_cap) { free(h); snprintf(c->err, sizeof(c->err), "shard window too small"); return QZD_ERR_DSTCAP; }
            hipError_t e = hipMemcpyAsync(pay + comp, s->slots[r], h[r].comp_len, hipMemcpyDeviceToDevice, c->st[0]);
            if (e != hipSuccess) { free(h); snprintf(c->err, sizeof(c->err), "shard root: slot %u -> member: %s", r, hipGetErrorString(e)); return QZD_ERR_HIP; }
        }
        raw += h[r].raw_len; comp += h[r].comp_len;
    }
    free(h);
    if (raw > 0xffffffffull || comp > 0xffffffffull) { snprintf(c->err, sizeof(c->err), "a gzip-ext member holds less than 4 GiB"); return QZD_ERR_PARAM; }
    unsigned char hdr[24], tr[8];
    member_frame(hdr, tr, level, raw, comp, crc);
    HIPCHK(c, hipMemcpyAsync(pay - QZD_SHARD_HDR, hdr, 24, hipMemcpyHostToDevice, c->st[0]));
    HIPCHK(c, hipMemcpyAsync(pay + comp, tr, 8, hipMemcpyHostToDevice, c->st[0]));
    HIPCHK(c, hipStreamSynchronize(c->st[0]));                      /* the slots are free again when this returns */
    *d_stream = pay - QZD_SHARD_HDR; *stream_len = QZD_SHARD_HDR + comp + 8;
    if (crc_out) *crc_out = crc;
    if (raw_total) *raw_total = raw;
    return QZD_OK;
}


/* ------------------------------------------------------------------ the same gather over RCCL
 *
 * north_star names RCCL over xGMI for "the trivial gather"; this is it, next to the IPC window above: the 32-byte records
 * travel as one ncclAllGather, the variable-size shards as one group of ncclSend (every rank but the root) / ncclRecv (the
 * root, one per rank, each at the offset the records give) on the context's stream - the waits are RCCL's, on the
 * device, no host polling.  The library is looked up at run time (librccl.so.1 of the ROCm the process runs on): a
 * single-GPU user of libqatzip_amd.so never loads half a gigabyte of collectives.  bench.py measures both transports
 * and reports the faster one; a box where RCCL cannot start (ranks sharing a device, no peer access) keeps the window. */
#include <dlfcn.h>
/* The library is looked up at run time, so its header is no reason for libqatzip_amd.so not to build: where
 * <rccl/rccl.h> is missing (a ROCm install without the development package) the few types and enumerators the calls
 * below need are stated here, as the ABI has them. */
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

struct qzd_rccl_api {
    void *so;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommAbort)(ncclComm_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    char why[200];                  /* why the library is not there (dlerror() at the time, kept: the call clears it) */
};
static qzd_rccl_api g_rccl;
static pthread_once_t g_rccl_once = PTHREAD_ONCE_INIT;
static void rccl_load(void)
{
    void *so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) {
        const char *e = dlerror();
        snprintf(g_rccl.why, sizeof(g_rccl.why), "%s", e ? e : "dlopen(librccl.so.1) failed");
        so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    }
    if (!so) return;
#define QZD_SYM(f) *(void **)&g_rccl.f = dlsym(so, "nccl" #f)
    QZD_SYM(GetUniqueId); QZD_SYM(CommInitRank); QZD_SYM(CommDestroy); QZD_SYM(CommAbort); QZD_SYM(GroupStart); QZD_SYM(GroupEnd);
    QZD_SYM(Send); QZD_SYM(Recv); QZD_SYM(AllGather); QZD_SYM(GetErrorString);
#undef QZD_SYM
    if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.GroupStart && g_rccl.GroupEnd &&
        g_rccl.Send && g_rccl.Recv && g_rccl.AllGather && g_rccl.GetErrorString) g_rccl.so = so;
    else snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks an entry point this library calls");
}
static bool rccl_ready(void) { pthread_once(&g_rccl_once, rccl_load); return g_rccl.so != NULL; }

struct qzd_rccl {
    qzd_ctx *ctx;
    uint32_t rank, world;
    uint64_t cap;
    ncclComm_t comm;
    uint8_t *win;                   /* root: [24-byte header | payload cap | 8-byte trailer] */
    qzd_shard_rec *d_recs;          /* world + 1 records: [0..world) gathered, [world] mine */
    qzd_shard_rec *h_recs;          /* pinned mirror */
    double timeout_s;               /* how long a wait for the other ranks may take (QATZIP_AMD_RCCL_TIMEOUT, default 60) */
};
#define RCCLCHK(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    snprintf((c)->err, sizeof((c)->err), "%s -> %s", #call, g_rccl.GetErrorString(r_)); return QZD_ERR_HIP; } } while (0)
/* inside ncclGroupStart / ncclGroupEnd: the group is closed on the way out, whatever happened in it */
#define RCCLCHK_G(c, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    snprintf((c)->err, sizeof((c)->err), "%s -> %s", #call, g_rccl.GetErrorString(r_)); g_rccl.GroupEnd(); return QZD_ERR_HIP; } } while (0)

/* RCCL's waits are on the device: a rank that never arrives leaves the others' kernels spinning and a plain
 * hipStreamSynchronize() waiting with them for ever.  The host polls the stream instead, and when the time is up the
 * communicator is aborted (the kernels leave) and the call fails everywhere it was waiting. */
static int rccl_wait(qzd_rccl *s, hipStream_t st, const char *what)
{
    qzd_ctx *c = s->ctx;
    const double t0 = now_s();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return QZD_OK;
        if (e != hipErrorNotReady) { snprintf(c->err, sizeof(c->err), "RCCL gather (%s): %s", what, hipGetErrorString(e)); return QZD_ERR_HIP; }
        if (now_s() - t0 > s->timeout_s) {
            snprintf(c->err, sizeof(c->err), "RCCL gather: rank %u waited %.0f s in %s for a rank that did not come; communicator aborted",
                     s->rank, s->timeout_s, what);
            if (s->comm && g_rccl.CommAbort) { g_rccl.CommAbort(s->comm); s->comm = NULL; }
            return QZD_ERR_HIP;
        }
        struct timespec ts = {0, 50000}; nanosleep(&ts, NULL);
    }
}

extern "C" int qzd_rccl_unique_id(uint8_t id_out[128])
{
    if (!id_out) return QZD_ERR_PARAM;
    if (!rccl_ready()) return QZD_ERR_UNSUPPORTED;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the ABI");
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return QZD_ERR_HIP;
    memcpy(id_out, &id, 128);
    return QZD_OK;
}

extern "C" void qzd_rccl_close(qzd_rccl *s)
{
    if (!s) return;
    hipSetDevice(s->ctx->device);
    if (s->comm) { hipDeviceSynchronize(); g_rccl.CommDestroy(s->comm); }
    if (s->win) hipFree(s->win);
    if (s->d_recs) hipFree(s->d_recs);
    if (s->h_recs) hipHostFree(s->h_recs);
    delete s;
}

/* every rank, with the 128 bytes rank 0 got from qzd_rccl_unique_id(); cap_bytes = payload capacity of the root's buffer */
extern "C" int qzd_rccl_create(qzd_ctx *c, uint32_t rank, uint32_t world, const uint8_t id[128], uint64_t cap_bytes, qzd_rccl **out)
{
    if (!c || !out || !id || world == 0 || rank >= world || world > 4096) return QZD_ERR_PARAM;
    *out = NULL;
    if (!rccl_ready()) { snprintf(c->err, sizeof(c->err), "librccl.so.1 not usable: %s", g_rccl.why[0] ? g_rccl.why : "?"); return QZD_ERR_UNSUPPORTED; }
    hipSetDevice(c->device);
    qzd_rccl *s = new (std::nothrow) qzd_rccl();
    if (!s) return QZD_ERR_HIP;
    s->ctx = c; s->rank = rank; s->world = world; s->cap = cap_bytes; s->comm = NULL; s->win = NULL; s->d_recs = NULL; s->h_recs = NULL;
    const char *te = getenv("QATZIP_AMD_RCCL_TIMEOUT");
    s->timeout_s = te && atof(te) > 0 ? atof(te) : 60.0;
    if (hipMalloc(&s->d_recs, (size_t)(world + 1) * sizeof(qzd_shard_rec)) != hipSuccess ||
        hipHostMalloc((void **)&s->h_recs, (size_t)(world + 1) * sizeof(qzd_shard_rec), hipHostMallocDefault) != hipSuccess ||
        (rank == 0 && hipMalloc(&s->win, QZD_SHARD_HDR + cap_bytes + 8 + 256) != hipSuccess)) {
        snprintf(c->err, sizeof(c->err), "RCCL gather: out of memory");
        qzd_rccl_close(s); return QZD_ERR_HIP;
    }
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    ncclResult_t r = g_rccl.CommInitRank(&s->comm, (int)world, uid, (int)rank);
    if (r != ncclSuccess) {
        snprintf(c->err, sizeof(c->err), "ncclCommInitRank(%u of %u on device %d) -> %s", rank, world, c->device, g_rccl.GetErrorString(r));
        s->comm = NULL; qzd_rccl_close(s); return QZD_ERR_HIP;
    }
    *out = s;
    return QZD_OK;
}

/* every rank: my shard (raw-deflate stream of my chunk range, its CRC-32) joins the member in the root's HBM.  On the
 * root *d_stream / *stream_len describe the finished member (valid until the next gather or qzd_rccl_close); the other
 * ranks get NULL / 0.  Returns when this rank's part is done (its stream is synchronised).  Every decision that sends a
 * rank down a different path is taken from the gathered records, which all ranks hold alike - the root's capacity
 * travels in its record - so that no rank leaves while the others wait for it; waits are bounded (rccl_wait). */
extern "C" int qzd_rccl_gather(qzd_rccl *s, const uint8_t *d_comp, uint64_t comp_len, uint64_t raw_len, uint32_t crc32, int level,
                               uint8_t **d_stream, uint64_t *stream_len, uint32_t *crc_out, uint64_t *raw_total)
{
    if (!s || (comp_len && !d_comp)) return QZD_ERR_PARAM;
    qzd_ctx *c = s->ctx;
    if (!s->comm) { snprintf(c->err, sizeof(c->err), "RCCL gather: the communicator was aborted by an earlier timeout"); return QZD_ERR_HIP; }
    hipSetDevice(c->device);
    hipStream_t st = c->st[0];
    qzd_shard_rec *mine = s->h_recs + s->world;
    mine->raw_len = raw_len; mine->comp_len = comp_len; mine->crc = crc32; mine->seq = 1; mine->done = 0;
    mine->pad = (uint32_t)(s->cap >> 12);                           /* my buffer's capacity in 4 KiB units: the root's is what counts */
    HIPCHK(c, hipMemcpyAsync(s->d_recs + s->world, mine, sizeof(*mine), hipMemcpyHostToDevice, st));
    RCCLCHK(c, g_rccl.AllGather(s->d_recs + s->world, s->d_recs, sizeof(qzd_shard_rec), ncclUint8, s->comm, st));
    HIPCHK(c, hipMemcpyAsync(s->h_recs, s->d_recs, (size_t)s->world * sizeof(qzd_shard_rec), hipMemcpyDeviceToHost, st));
    int rc = rccl_wait(s, st, "the all-gather of the records");
    if (rc) return rc;
    uint64_t raw = 0, comp = 0; uint32_t crc = 0;
    for (uint32_t r = 0; r < s->world; r++) {
        crc = r == 0 ? s->h_recs[r].crc : qzd_crc32_combine(crc, s->h_recs[r].crc, s->h_recs[r].raw_len);
        raw += s->h_recs[r].raw_len; comp += s->h_recs[r].comp_len;
    }
    /* the same verdict on every rank: nobody posts a send the root will not receive */
    if (raw > 0xffffffffull || comp > 0xffffffffull) { snprintf(c->err, sizeof(c->err), "a gzip-ext member holds less than 4 GiB"); return QZD_ERR_PARAM; }
    if (comp > ((uint64_t)s->h_recs[0].pad << 12)) { snprintf(c->err, sizeof(c->err), "RCCL gather: root buffer too small"); return QZD_ERR_DSTCAP; }
    if (s->rank == 0) {
        uint8_t *pay = s->win + QZD_SHARD_HDR;
        if (comp_len) HIPCHK(c, hipMemcpyAsync(pay, d_comp, comp_len, hipMemcpyDeviceToDevice, st));
        RCCLCHK(c, g_rccl.GroupStart());
        uint64_t off = s->h_recs[0].comp_len;
        for (uint32_t r = 1; r < s->world; r++) {
            if (s->h_recs[r].comp_len) RCCLCHK_G(c, g_rccl.Recv(pay + off, s->h_recs[r].comp_len, ncclUint8, (int)r, s->comm, st));
            off += s->h_recs[r].comp_len;
        }
        RCCLCHK(c, g_rccl.GroupEnd());
        unsigned char *ht = (unsigned char *)(s->h_recs + s->world);     /* pinned: 32 bytes = header 24 + trailer 8 */
        member_frame(ht, ht + 24, level, raw, comp, crc);
        HIPCHK(c, hipMemcpyAsync(s->win, ht, 24, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(pay + comp, ht + 24, 8, hipMemcpyHostToDevice, st));
        rc = rccl_wait(s, st, "the receives");
        if (rc) return rc;
        if (d_stream) *d_stream = s->win;
        if (stream_len) *stream_len = QZD_SHARD_HDR + comp + 8;
    } else {
        if (comp_len) {
            RCCLCHK(c, g_rccl.GroupStart());
            RCCLCHK_G(c, g_rccl.Send(d_comp, comp_len, ncclUint8, 0, s->comm, st));
            RCCLCHK(c, g_rccl.GroupEnd());
        }
        rc = rccl_wait(s, st, "the send");
        if (rc) return rc;
        if (d_stream) *d_stream = NULL;
        if (stream_len) *stream_len = 0;
    }
    if (crc_out) *crc_out = crc;
    if (raw_total) *raw_total = raw;
    return QZD_OK;
}
