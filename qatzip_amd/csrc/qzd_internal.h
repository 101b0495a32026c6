/* qzd_internal.h — context shared by the host-side translation units of libqatzip_amd.so */
#ifndef QZD_INTERNAL_H
#define QZD_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <pthread.h>
#include "../../include/qzamd_device.h"
#include "qzk_deflate_lz77.h"

/* chunks per batch = QZD_BATCH_ROUNDS chunks for every resident K1 workgroup (qzd_ctx::batch_chunks): a batch whose size
 * is not a multiple of the workgroup count leaves part of the chip idle during its last round */
#define QZD_BATCH_ROUNDS 3u
#define QZD_NBUF 2
#define QZD_K1EV 64                  /* K1 launches per call that get their own pair of timing events */
#define QZD_K1_WGS_PER_CU ((uint32_t)(QZK_K1_WAVES * QZK_K1_OCC))   /* pulling waves per CU */

struct qzd_ctx {
    int device;
    hipStream_t st[QZD_NBUF];
    hipStream_t st_copy; hipEvent_t cp_ev[QZD_NBUF + 1];   /* host input arrives batch by batch while the previous batch is parsed */
    hipStream_t st_out;                                     /* decoded output on its way to the host */
    hipStream_t pq_copy, pq_out;                            /* the same two for a piece-wise decode (qzd_inflate_stream_from_host): streams
                                                             * with hardware queues of their own, made by the first such call
                                                             * (qzd_device.hip, stream_own_queue / qzd_pipe_streams) */
    bool helper;                                            /* a piece's helper context (qzd_inflate_stream_from_host): one stream */
    hipEvent_t done[QZD_NBUF], k1done[QZD_NBUF];
    uint32_t cus;                                   /* compute units of the device */
    /* output slots of the LZ4 frame kernel (the deflate pipeline's scratch lives in the device's pool, qzd_k1pool) */
    uint8_t *slots[QZD_NBUF];
    /* K1 (persistent pull kernel): one candidate table (65536 x QZK_K1_WAVES entries of 16 bytes = 4 MiB) per resident
     * workgroup, one chunk counter per buffer set */
    /* The tables belong to the DEVICE, not to the context (qzd_k1pool): every session of a process on one GPU parses
     * with the same 5 GiB - a context borrows them from its first K1 launch of a call until qzd_sync(). */
    uint32_t *k1_counter;
    uint32_t k1_wgs;                                /* resident pulling WAVES (QZK_K1_WAVES per workgroup, QZK_K1_OCC workgroups per CU) */
    uint32_t batch_chunks;
    /* K1 launch durations (HIP events around every K1 launch, harvested at qzd_sync): bench.py's roofline input */
    hipEvent_t k1ev[QZD_K1EV][2]; uint32_t k1ev_chunks[QZD_K1EV]; uint32_t k1ev_n;
    double k1_ms_acc; uint64_t k1_launch_acc, k1_chunk_acc;
    size_t slot_cap;
    /* per-call arrays */
    uint32_t *d_len, *d_crc; uint64_t *d_offs; uint32_t call_cap;
    uint64_t *d_running; uint32_t *d_overflow;
    uint64_t *h_running; uint32_t *h_overflow;      /* pinned */
    uint8_t *d_lz4tab; uint32_t lz4tab_waves;       /* qzk_lz4c_pull_kernel: frame counter (256 B) + one 16 KiB hash table per resident wave */
    uint32_t *h_wm;                                 /* pinned, read by qzk_lz77_pull_kernel: [0] chunks of host input landed, [1] a wave gave up waiting */
    /* timing */
    hipEvent_t ev[QZD_NBUF][4]; hipEvent_t ev_begin, ev_end;
    uint32_t nbatches; float ms[4];
    uint32_t last_nchunks;
    /* generic small scratch for the decompress side: device + pinned host mirror */
    uint8_t *d_aux, *h_aux; size_t aux_cap;
    /* output streaming (qzd_inflate_stream_to_host): while set, the two-phase decoder resolves the output range by range
     * and sends each range to so_host behind its launch; so_nat[i] = array index of the segment that is i-th in the output */
    uint8_t *so_host; const uint32_t *so_nat; uint64_t so_sent; hipEvent_t so_ev[8];
    uint8_t *d_big; size_t big_cap; uint32_t big_small;   /* device-only scratch (per-segment decode tables of K3b); calls in a row that needed under a quarter of it */
    uint32_t *d_cdesc; uint32_t cdesc_cap;          /* per-slot descriptors of a coalesced launch (qzd_deflate_slots) */
    uint8_t *d_lane; size_t lane_cap;               /* device-only scratch of the one-chunk-per-lane compress path (K1b) */
    float inf_ms[4];
    bool no_stream_in;              /* this call is the batched retry of a launch that gave up waiting for its input */
    /* device arrays of the last two-phase inflate (phase A done, phase B still to run): two_phase() / two_phase_resolve() */
    struct { void *segs, *res, *ts, *lits, *seqs, *chains; uint32_t *ord; uint32_t nsegs, K; } tp;
    /* helpers of a piece-wise host-to-host decode (qzd_inflate_stream_from_host): contexts of their own - streams, scratch,
     * pinned staging - so that the pieces' phases run side by side; made at the first such call, freed with this context */
    struct qzd_ctx *pipe_ctx[8]; hipEvent_t pipe_ev[8];
    uint32_t own_queues;                            /* streams of this context that hold a hardware queue of their own (counted per device) */
    char err[256];
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    return QZD_ERR_HIP; } } while (0)

/* K1's candidate tables, one set per device and process: 65536 x QZK_K1_WAVES entries of 16 bytes per workgroup, grown
 * on demand; `epoch` = next unused chunk epoch (entries are tagged with it; never 0).  The calls of different contexts
 * use the pool one after the other ON THE GPU: every call's streams first wait for `busy` (recorded behind the previous
 * call's last launch), so no host thread ever holds anything across API calls; `lock` only guards these fields while a
 * call is being enqueued (one big call fills the chip anyway; small calls meet in the coalescing queue, not here). */
struct qzd_k1pool {
    pthread_mutex_t lock; qzk_bkt *tables; uint32_t tab_wgs; uint32_t epoch;
    hipEvent_t busy; bool busy_valid;
    /* K1 -> K2 hand-over of a batch (symbols, block marks) and K2's output slots, two sets: the same lifetime as the
     * tables' loan, so they are the device's as well (6.3 GiB at 64 KB chunks, once instead of per session) */
    uint8_t *sym_lc[2]; uint16_t *sym_dist[2]; uint8_t *slots[2]; qzk_lzmeta *meta[2];
    size_t sym_cap, slot_cap; uint32_t meta_cap;
};
#define QZD_MAX_DEVICES 64

/* grow the aux scratch pair to at least n bytes */
int qzd_aux_reserve(qzd_ctx *c, size_t n);
int qzd_create_helper(int device, qzd_ctx **out);
int qzd_pipe_streams(qzd_ctx *c);

/* host-side CRC-32 helpers (zlib crc32_combine semantics) */
extern "C" uint32_t qzd_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

#endif
