/*
 * qzk_checksum.h — K6 standalone: CRC-32 of arbitrary byte ranges already resident in
 * HBM (one range per 256-thread workgroup), used to verify gzip trailers after K3
 * (decompOutCheckSum, src/qatzip_utils.c:1483-1532) and by the LZ4 path.
 * The workgroup CRC routine itself lives in qzk_deflate_huff.h (qzk_block_crc32):
 * per-thread table-driven CRC of a contiguous slice, then x^(8*tail) mod P shifts
 * and an XOR reduction — crc32_combine() semantics without any serial pass.
 */
#ifndef QZK_CHECKSUM_H
#define QZK_CHECKSUM_H
#include "qzk_deflate_huff.h"

typedef struct { uint64_t off; uint32_t len; uint32_t pad; } qzk_range;

QZ_KERNEL qzk_crc_kernel(const uint8_t *data, const qzk_range *ranges, uint32_t nranges, uint32_t *crc_out)
{
    QZ_LDS qzk_crc_lds S;
    const uint32_t r = blockIdx.x;
    if (r >= nranges) return;
    const uint32_t c = qzk_block_crc32(&S, data + ranges[r].off, ranges[r].len);
    if (threadIdx.x == 0) crc_out[r] = c;
}

/* CRC-32 of every hw_buff_sz chunk of a buffer (the per-chunk values the gzip trailers are folded from,
 * src/qatzip_sw.c:219-231 keeps them as zlib's running crc) */
QZ_KERNEL qzk_crc_chunks_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks, uint32_t *crc_out)
{
    QZ_LDS qzk_crc_lds S;
    const uint32_t c = blockIdx.x;
    if (c >= nchunks) return;
    const uint64_t off = (uint64_t)c * chunk_sz;
    const uint32_t n = (uint32_t)((src_len - off) < chunk_sz ? (src_len - off) : chunk_sz);
    const uint32_t v = qzk_block_crc32(&S, src + off, n);
    if (threadIdx.x == 0) crc_out[c] = v;
}

#endif
