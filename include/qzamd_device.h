/*
 * qzamd_device.h — device-resident C ABI of the MI355X backend (libqatzip_amd.so).
 *
 * This is the layer that takes the place of the QAT driver calls on the
 * reference's hot path: where src/qatzip.c:1542 submits one hw_buff_sz chunk with
 * cpaDcCompressData2() and src/qatzip.c:1633 polls it back, the MI355X backend
 * submits ALL chunks of a call as one batch of workgroups.  Pointers prefixed d_
 * are device (HBM) pointers, h_ are host pointers; sizes are plain integers; no
 * torch / C++ types cross this boundary.  The qatzip.h API (include/qatzip.h) is
 * implemented on top of it in qatzip_amd/csrc/qz_api.cpp; bench.py and the GPU
 * parity tests also call it directly with buffers already resident in HBM.
 *
 * All functions return 0 on success or a negative QZD_* code.  They are
 * synchronous with respect to the host (they return after the work finished)
 * unless stated otherwise.
 */
#ifndef QZAMD_DEVICE_H
#define QZAMD_DEVICE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define QZD_OK 0
#define QZD_ERR_PARAM (-1)
#define QZD_ERR_HIP (-2)
#define QZD_ERR_DSTCAP (-3)      /* destination too small (maps to QZ_BUF_ERROR / QZ_FAIL) */
#define QZD_ERR_DATA (-4)        /* corrupt compressed input (maps to QZ_DATA_ERROR) */
#define QZD_ERR_UNSUPPORTED (-5)
#define QZD_ERR_NOMEM (-6)       /* device memory for a call's scratch could not be had (maps to QZ_NOSW_LOW_MEM) */

typedef struct qzd_ctx qzd_ctx;

/* one context per (host thread, GPU); owns streams + scratch HBM */
int qzd_create(int device, qzd_ctx **ctx);
void qzd_destroy(qzd_ctx *ctx);
const char *qzd_last_error(qzd_ctx *ctx);
int qzd_device_count(void);
int qzd_ctx_device(qzd_ctx *ctx);      /* the GPU this context was created on (-1: no context) */
/* chunks the compress path hands to one launch of its LZ77 kernel on this device (a whole number of rounds over
 * the resident workgroups); callers that pipeline their own work can size it in these units */
uint32_t qzd_batch_chunks(qzd_ctx *ctx);
/* accumulated duration (HIP events on the launching stream), count and chunk total of the LZ77 kernel launches
 * since the last reset; harvested whenever qzd_sync() completes a compress call */
int qzd_k1_stats(qzd_ctx *ctx, double *ms, uint64_t *launches, uint64_t *chunks, int reset);

/* measured stream-copy rate of the device, read + written decimal GB per second (a hand-written 16-byte-per-lane copy
 * kernel over two buffers of `bytes` each, best of `iters`): the yardstick beside the 8 TB/s datasheet figure */
int qzd_stream_copy_peak(qzd_ctx *ctx, uint64_t bytes, int iters, double *gbps);
/* what the host link delivers: pinned hipMemcpyAsync of `bytes` each way, best of `iters`, decimal GB/s */
int qzd_pcie_peak(qzd_ctx *ctx, uint64_t bytes, int iters, double *h2d_gbps, double *d2h_gbps);

/* plain HBM / pinned-host memory helpers (replace qaeMemAllocNUMA, src/qatzip_mem.c:169-224) */
void *qzd_dev_alloc(qzd_ctx *ctx, size_t n);
void qzd_dev_free(qzd_ctx *ctx, void *d_p);
int qzd_h2d(qzd_ctx *ctx, void *d_dst, const void *h_src, size_t n);
int qzd_d2h(qzd_ctx *ctx, void *h_dst, const void *d_src, size_t n);
int qzd_d2d(qzd_ctx *ctx, void *d_dst, const void *d_src, size_t n);      /* on the context's copy stream; returns when done */
void *qzd_host_alloc_pinned(size_t n);
void qzd_host_free_pinned(void *p);

/*
 * Raw-deflate `n` bytes at d_src exactly as the reference's software path does
 * for one qzCompress() call (src/qatzip_sw.c:178-231): independent chunks of
 * chunk_sz bytes, zlib level `level` (1-9; 1 is the tuned path), every chunk closed by
 * the Z_FULL_FLUSH marker except — when last != 0 — the final one, which
 * carries BFINAL.  n == 0 emits the single empty final block (last) or the
 * bare marker.  The stream is written contiguously to d_dst.
 *   h_out_len    <- stream length
 *   h_chunk_crc  <- (optional) CRC-32 of every input chunk, nchunks entries
 * d_src must be readable for n bytes; d_dst needs dst_cap bytes.
 */
int qzd_deflate_raw(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level, int last,
                    uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_chunk_crc);

/* qzd_deflate_raw for input that is still in host memory (what qzCompress is handed): h_src travels to d_stage (n bytes
 * of device memory the caller provides) one batch at a time, each copy issued behind the previous batch's kernels, so
 * PCIe time hides under the parse instead of preceding it */
int qzd_deflate_raw_from_host(qzd_ctx *ctx, const uint8_t *h_src, uint8_t *d_stage, uint64_t n, uint32_t chunk_sz, int level,
                              int last, uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_chunk_crc);

/*
 * Many small requests in ONE launch — what the reference's asynchronous API is for (qzCompress2's ring + consumer
 * thread, src/qatzip.c:3103-4110, keeps many requests in flight on the accelerator).  d_src holds nslots slots of
 * chunk_sz bytes; a request occupies consecutive slots starting on a slot boundary; h_cdesc[k] = number of bytes in
 * slot k (<= chunk_sz), with bit 31 set on the slot that ends its request: that slot's stream carries BFINAL, the
 * others end in the flush marker, exactly as separate qzd_deflate_raw(last = 1) calls would write them.  All slot
 * streams are written back to back to d_dst; h_slot_len[k] / h_slot_crc[k] give each slot's share and CRC-32.
 */
int qzd_deflate_slots(qzd_ctx *ctx, const uint8_t *d_src, uint32_t nslots, uint32_t chunk_sz, int level,
                      const uint32_t *h_cdesc, uint8_t *d_dst, uint64_t dst_cap, uint64_t *h_out_len,
                      uint32_t *h_slot_len, uint32_t *h_slot_crc);

/* How the qzCompress2 submission queue has been doing: launches that carried more than one request, and the requests
 * they carried (process-wide, since load).  Exported by the same library; not part of qatzip.h. */
void qzamd_async_stats(uint64_t *launches, uint64_t *requests);

/* Which of the reference's two wire framings the compress side of a qatzip.h session writes (QzSession_T from
 * include/qatzip.h): 0 = the software path's (one member per stream, a full-flush marker after every hw_buff_sz chunk,
 * src/qatzip_sw.c:178-253 - the default, and what the bit-exactness claim is about); 1 = the hardware path's (one complete
 * member per chunk: qzGzipHeaderGen / stdGzipHeaderGen / qz4BHeaderGen + footer, src/qatzip.c:1691-1718,
 * src/qatzip_gzip.c:86-143), for interop with consumers of QAT-produced streams.  QATZIP_AMD_HW_FRAMING=1 makes it the
 * default of new sessions.  Not callable in the middle of a stream opened with last = 0. */
struct QzSession_S;
int qzamd_set_hw_framing(struct QzSession_S *sess, int on);

/* asynchronous flavour used by bench.py: enqueue only; qzd_sync() + qzd_result() finish it */
int qzd_deflate_raw_async(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint32_t chunk_sz, int level, int last,
                          uint8_t *d_dst, uint64_t dst_cap);
int qzd_sync(qzd_ctx *ctx);
int qzd_result(qzd_ctx *ctx, uint64_t *h_out_len, uint32_t *h_chunk_crc, uint32_t nchunks);

/* compressed length of each chunk of the last deflate call (nchunks entries) */
int qzd_chunk_lens(qzd_ctx *ctx, uint32_t *h_len, uint32_t nchunks);

/* elapsed GPU time (ms) of the kernels of the last *_async call, per kernel family, measured
 * with hipEvents on the stream the kernels ran on: [0]=lz77 [1]=huffman [2]=scan+gather [3]=total */
int qzd_last_timing(qzd_ctx *ctx, float ms[4]);

/* ------------------------------------------------------------------ decompress side
 *
 * Segment records shared with the kernels (see qatzip_amd/csrc/qzk_inflate.h):
 *   qzd_infseg { u64 in_off; u64 out_off; u32 in_len; u32 out_cap; u32 flags; u32 pad; }
 *     pad: optional hint, the compressed length of the segment when in_len is only an upper bound (0 = no hint);
 *     with a hint the two-phase path decodes a segment with several lanes (speculative sub-segment decoding)
 *   qzd_infres { i32 status; u32 in_used; u32 out_len; u32 nblocks; }
 * status: 0 = ended with BFINAL, 1 = ended at a flush marker, <0 = error
 * (-1 data, -2 output capacity, -3 input exhausted, -4 needs earlier history).
 * flags: 1 = count only (write nothing), 2 = continue through flush markers.
 */
typedef struct { uint64_t in_off, out_off; uint32_t in_len, out_cap, flags, pad; } qzd_infseg;
typedef struct { int32_t status; uint32_t in_used, out_len, nblocks; } qzd_infres;
typedef struct { uint64_t off; uint32_t len, pad; } qzd_range;

/* raw-inflate nsegs independent segments (one wave each); what qzDecompress does per member when the
 * member sizes are known from the gzip-ext header (src/qatzip_utils.c:1232-1345) */
int qzd_inflate_segments(qzd_ctx *ctx, const uint8_t *d_comp, uint8_t *d_out, const void *h_segs,
                         uint32_t nsegs, void *h_res);

/* raw-inflate ONE deflate stream that starts at d_src and ends with its BFINAL block; the stream is cut
 * at Z_FULL_FLUSH markers and decoded segment-parallel (seg_hint = expected bytes per segment, normally
 * the session's hw_buff_sz; 0 = unknown).  Replaces the inflate() loop of src/qatzip_sw.c:339.
 *   h_in_used <- compressed bytes consumed   h_out_len <- bytes produced
 *   h_crc     <- (optional) CRC-32 of the output, for the gzip trailer check */
int qzd_inflate_stream(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                       uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc);

/* qzd_inflate_stream for a caller that wants the output in host memory (what qzDecompress is asked for): when the
 * stream decodes in the first, optimistic pass of a large call, the output is sent to h_dst range by range behind the
 * kernels that produce it, and *h_sent = 1; otherwise *h_sent = 0 and the output is (only) in d_dst */
int qzd_inflate_stream_to_host(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                               uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc,
                               uint8_t *h_dst, int *h_sent);

/* The same for a member that is still in HOST memory (qzDecompress, src/qatzip.c:2103-2404 keeps its engine fed while
 * earlier requests retire): the n bytes at h_src go to d_src piece by piece, each piece is decoded as soon as it has
 * landed and its output leaves for h_dst while the later pieces are still arriving and decoding.  When it returns all n
 * bytes stand at d_src, whatever the result.  A stream the pieces cannot take goes through qzd_inflate_stream_to_host
 * from there (same results, same error codes).  QATZIP_AMD_PIPE=<pieces> overrides the piece count (0: never). */
int qzd_inflate_stream_from_host(qzd_ctx *ctx, const uint8_t *h_src, uint64_t n, uint8_t *d_src, uint8_t *d_dst, uint64_t dst_cap,
                                 uint32_t seg_hint, uint64_t *h_in_used, uint64_t *h_out_len, uint32_t *h_crc,
                                 uint8_t *h_dst, int *h_sent);

/* Adler-32 (zlib adler32(), what the DEFLATE_ZLIB trailer carries: deflateInit2 with windowBits 15,
 * src/qatzip_sw.c:147) of every chunk_sz chunk of HBM-resident data; fold with qzd_adler32_combine */
int qzd_adler32_chunks(qzd_ctx *ctx, const uint8_t *d_data, uint64_t n, uint32_t chunk_sz, uint32_t *h_adler);
uint32_t qzd_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2);

/* CRC-32 (zlib crc32()) of HBM-resident data */
int qzd_crc32(qzd_ctx *ctx, const uint8_t *d_data, uint64_t n, uint32_t *h_crc);
int qzd_crc32_ranges(qzd_ctx *ctx, const uint8_t *d_data, const void *h_ranges, uint32_t nranges, uint32_t *h_crc);

/* ------------------------------------------------------------------ LZ4 frames
 * qzd_lz4seg { u64 in_off; u64 out_off; u32 in_len; u32 out_cap; }
 * qzd_lz4res { i32 status; u32 in_used; u32 out_len; u32 pad; }   status 0 ok, -1 data, -2 capacity, -3 truncated */
typedef struct { uint64_t in_off, out_off; uint32_t in_len, out_cap; } qzd_lz4seg;
typedef struct { int32_t status; uint32_t in_used, out_len, pad; } qzd_lz4res;

/* every frame_sz (<= 64 KB) bytes of d_src -> one LZ4 frame exactly as LZ4F_compressFrame emits it for the
 * preferences of src/qatzip_sw.c:451-456 (content size + content checksum, one independent block, stored
 * when incompressible) when it is called on that piece alone; frames are written back to back.  h_frame_len (optional) <- size of every frame. */
int qzd_lz4_compress_frames(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint32_t frame_sz, uint8_t *d_dst,
                            uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_frame_len);
/* the same frames with the header the reference's HARDWARE path writes per chunk (qzLZ4HeaderGen,
 * src/qatzip_lz4.c:104-132: FLG 0x4C, content size = the chunk's bytes); footer as qzLZ4FooterGen (:134-143).
 * frame_sz above 64 KB (a hw_buff_sz of 128 KB .. 512 KB): every chunk's frame holds linked 64 KB blocks - one wave per
 * chunk, all chunks of the call in one launch */
int qzd_lz4_compress_frames_hw(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint32_t frame_sz, uint8_t *d_dst,
                               uint64_t dst_cap, uint64_t *h_out_len, uint32_t *h_frame_len);
/* ONE call above 64 KB as the one frame LZ4F_compressFrame writes for it (src/qatzip_sw.c:451-456): FLG 0x4C, the 64 KB
 * blocks linked - one parse state for the whole frame, hence one wave's serial work; 64 KB < n <= 0x7fff0000 (beyond that
 * liblz4 rescales its 32-bit positions, which is not reproduced) */
int qzd_lz4_compress_linked(qzd_ctx *ctx, const uint8_t *d_src, uint64_t n, uint8_t *d_dst, uint64_t dst_cap,
                            uint64_t *h_out_len);
/* decode nsegs frames (any block mode, content checksum verified on the GPU); replaces LZ4F_decompress,
 * src/qatzip_sw.c:496 */
int qzd_lz4_decompress_frames(qzd_ctx *ctx, const uint8_t *d_comp, uint8_t *d_out, const void *h_segs,
                              uint32_t nsegs, void *h_res);

/* ------------------------------------------------------------------ one member from several GPUs
 *
 * What is left of the reference's in-order retire (doCompressOut: payload copy + crc32_combine + footer,
 * src/qatzip.c:1691-1718) when the chunks of ONE logical buffer are deflated by several GPUs, one process each: rank r
 * holds the raw-deflate stream of its contiguous chunk range (qzd_deflate_raw with last = 0, the last rank last = 1) and
 * its CRC-32.  The root (rank 0) owns a window in its HBM that every rank maps through HIP IPC; qzd_shard_put() publishes
 * a 32-byte record {raw bytes, compressed bytes, CRC-32}, derives the shard's offset from the records of the ranks before
 * it and copies the shard there - a peer-to-peer write over xGMI, no host bounce, no collective;  qzd_shard_finish() on
 * the root waits for all shards, folds the CRCs (crc32_combine) and closes the member: the 24-byte gzip-ext header with
 * both sizes in front, CRC-32 + ISIZE behind - byte for byte what one qzCompress() call over the whole buffer writes.
 * The launcher only has to hand the root's 64-byte handle to the other ranks (any transport; bench.py uses gloo).
 * seq != 0 names the stream; the window is reused for the next one under the next number.  Waits give up after
 * timeout_s seconds. */
typedef struct qzd_shard qzd_shard;
int qzd_shard_root_create(qzd_ctx *ctx, uint32_t world, uint64_t cap_bytes, uint8_t handle_out[64], qzd_shard **out);
int qzd_shard_attach(qzd_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t handle[64], uint64_t cap_bytes, qzd_shard **out);
/* round 5: the compressed shards travel through one SLOT per non-root rank (an allocation of its own, mapped by that rank
 * alone) instead of one window of world x shard bytes - hipIpcOpenMemHandle never came back for a window above 2 GiB.
 * Root: qzd_shard_slot_handle(root, r, handle) for every rank r >= 1; rank r: qzd_shard_attach_slot(mine, that handle)
 * before its first qzd_shard_put(). */
int qzd_shard_slot_handle(qzd_shard *root, uint32_t rank, uint8_t handle_out[64]);
int qzd_shard_attach_slot(qzd_shard *shard, const uint8_t handle[64]);
int qzd_shard_put(qzd_shard *s, const uint8_t *d_comp, uint64_t comp_len, uint64_t raw_len, uint32_t crc32, uint32_t seq,
                  double timeout_s, uint64_t *h_offset);
/* level: the comp_lvl the shards were deflated at (the header's XFL byte follows it, as in a qzCompress call) */
int qzd_shard_finish(qzd_shard *s, uint32_t seq, double timeout_s, int level, uint8_t **d_stream, uint64_t *stream_len,
                     uint32_t *crc_out, uint64_t *raw_total);
void qzd_shard_close(qzd_shard *s);
/* The same gather over RCCL (the collective library over xGMI that north_star names): the records as one ncclAllGather,
 * the shards as one group of ncclSend / ncclRecv into the root's HBM, every wait on the device.  Rank 0 makes the
 * 128-byte id, the launcher hands it to every rank (any transport), every rank creates its end with it.  librccl.so.1 is
 * looked up at run time; QZD_ERR_UNSUPPORTED when it is not there.  qzd_rccl_gather(): every rank passes its shard; on
 * the root *d_stream / *stream_len describe the finished member, byte for byte what qzd_shard_finish() builds. */
typedef struct qzd_rccl qzd_rccl;
int qzd_rccl_unique_id(uint8_t id_out[128]);
int qzd_rccl_create(qzd_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t id[128], uint64_t cap_bytes, qzd_rccl **out);
int qzd_rccl_gather(qzd_rccl *s, const uint8_t *d_comp, uint64_t comp_len, uint64_t raw_len, uint32_t crc32, int level,
                    uint8_t **d_stream, uint64_t *stream_len, uint32_t *crc_out, uint64_t *raw_total);
void qzd_rccl_close(qzd_rccl *s);
/* zlib crc32_combine(): CRC-32 of A || B from the CRC-32s of A and B and the length of B */
uint32_t qzd_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);
/* the same folded over the per-chunk CRC-32s of an n-byte buffer cut every chunk_sz bytes (host arrays) */
uint32_t qzd_crc32_fold(const uint32_t *h_crc, uint32_t nchunks, uint32_t chunk_sz, uint64_t n);

/* GPU time (ms) of the last qzd_inflate_stream call: [0] the inflate step, first launch to last (on the two-phase path the
 * host's walk through phase A's results lies inside it), [1] crc kernels, [2] phase B of the two-phase path (the
 * match-resolve kernel; 0 on the wave-per-segment path), [3] phase A (the Huffman-decoding kernel, with the launch for any
 * segment handed back to the one-lane kernel; 0 on the wave-per-segment path) */
int qzd_last_inflate_timing(qzd_ctx *ctx, float ms[4]);
/* device memory (bytes) the context's decode scratch holds at the moment - token sub-streams, marks, decode tables of the
 * two-phase inflate; it grows with the largest call and is given back after eight calls in a row that needed under a quarter */
uint64_t qzd_inflate_scratch_bytes(qzd_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
