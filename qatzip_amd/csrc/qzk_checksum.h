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
QZ_KERNEL qzk_crc_chunks_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks, uint32_t *crc_out,
                                const uint32_t *cdesc /* per-chunk lengths of a coalesced launch, or NULL */)
{
    QZ_LDS qzk_crc_lds S;
    const uint32_t c = blockIdx.x;
    if (c >= nchunks) return;
    const uint64_t off = (uint64_t)c * chunk_sz;
    const uint32_t n = cdesc ? (cdesc[c] & 0x7fffffffu) : (uint32_t)((src_len - off) < chunk_sz ? (src_len - off) : chunk_sz);
    const uint32_t v = qzk_block_crc32(&S, src + off, n);
    if (threadIdx.x == 0) crc_out[c] = v;
}

/* Adler-32 (zlib adler32(), the DEFLATE_ZLIB trailer: src/qatzip_sw.c:147 with windowBits 15) of every chunk.
 * 256 threads, each sums one contiguous slice (a = sum of bytes, b = sum of the running a); the slices fold in order
 * like adler32_combine(): a slice's bytes ride on the s1 reached before it, everything mod 65521. */
#define QZK_ADLER_MOD 65521u
QZ_KERNEL qzk_adler_chunks_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks, uint32_t *out)
{
    QZ_LDS uint32_t pa[QZK_HT], pb[QZK_HT];
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t off = (uint64_t)c * chunk_sz;
    const uint32_t n = (uint32_t)((src_len - off) < chunk_sz ? (src_len - off) : chunk_sz);
    const uint32_t S = (n + QZK_HT - 1) / QZK_HT;                 /* slice length, <= 2048 for a 512 KiB chunk */
    const uint32_t b0 = t * S < n ? t * S : n, b1 = (t + 1) * S < n ? (t + 1) * S : n;
    const uint8_t *p = src + off;
    uint32_t a = 0; uint64_t b = 0;
    for (uint32_t i = b0; i < b1; i++) { a += p[i]; b += a; }
    pa[t] = a % QZK_ADLER_MOD; pb[t] = (uint32_t)(b % QZK_ADLER_MOD);
    qz_block_sync();
    if (t == 0) {
        uint64_t A = 1, B = 0;                                     /* adler32 starts at s1 = 1, s2 = 0 */
        for (uint32_t k = 0; k < QZK_HT; k++) {
            const uint32_t k0 = k * S < n ? k * S : n, k1 = (k + 1) * S < n ? (k + 1) * S : n, len = k1 - k0;
            B = (B + (uint64_t)len * A + pb[k]) % QZK_ADLER_MOD;
            A = (A + pa[k]) % QZK_ADLER_MOD;
        }
        out[c] = (uint32_t)(B << 16 | A);
    }
}

#endif
