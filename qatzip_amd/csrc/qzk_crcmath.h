/*
 * qzk_crcmath.h — CRC-32 (reflected, polynomial 0xEDB88320) as arithmetic in GF(2)[x] mod P: what crc32_combine() does
 * (the software path folds its chunk CRCs with zlib's running crc32, src/qatzip_sw.c:219-231), shared by the workgroup
 * CRC routine (qzk_deflate_huff.h) and by the CRC that rides along K1's input reads (qzk_deflate_lz77.h).
 */
#ifndef QZK_CRCMATH_H
#define QZK_CRCMATH_H
#include "qzk_common.h"

#define QZK_POLY 0xEDB88320u
QZ_DEV uint32_t qzk_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ QZK_POLY : b >> 1;
    }
    return p;
}
/* x^(n * 2^k) mod P, using S->x2n[i] = x^(2^i) */
QZ_DEV uint32_t qzk_x2nmodp(const uint32_t *x2n, uint64_t n, unsigned k)
{
    uint32_t p = 1u << 31;
    while (n) { if (n & 1) p = qzk_multmodp(x2n[k & 31], p); n >>= 1; k++; }
    return p;
}

/* what a K1 workgroup keeps in LDS for the CRC of its waves' chunks, 1.1 KiB (nibble tables: K1 spends its LDS on waves):
 * adv[k][v] = the CRC register v << 4k advanced over four zero bytes (a dword's CRC is the XOR of eight of them),
 * mul[k][v] = (v << 4k) * x^(8*256) (a lane's consecutive dwords are 256 bytes apart), x2n[i] = x^(2^i) */
typedef struct { uint32_t adv[8][16]; uint32_t mul[8][16]; uint32_t x2n[32]; } qzk_k1crc_lds;

/* filled by the whole workgroup (any size >= 64), ends with a barrier */
QZ_DEV void qzk_k1crc_init(qzk_k1crc_lds *S)
{
    const uint32_t t0 = threadIdx.x, nt = blockDim.x;
    if (t0 == 0) {
        uint32_t p = 1u << 30;
        S->x2n[0] = p;
        for (int i = 1; i < 32; i++) S->x2n[i] = p = qzk_multmodp(p, p);
    }
    qz_block_sync();
    const uint32_t K = qzk_x2nmodp(S->x2n, 256, 3);
    for (uint32_t t = t0; t < 128; t += nt) {
        const uint32_t k = t >> 4, v = t & 15;
        uint32_t c = v << (4 * k);
        for (int i = 0; i < 32; i++) c = (c & 1) ? QZK_POLY ^ (c >> 1) : c >> 1;
        S->adv[k][v] = c;
        S->mul[k][v] = v ? qzk_multmodp(K, v << (4 * k)) : 0;
    }
    qz_block_sync();
}

/* acc * x^(8*256) + crc32(the four bytes of w): one Horner step of a lane over its column of the input */
QZ_DEV uint32_t qzk_k1crc_step(const qzk_k1crc_lds *S, uint32_t acc, uint32_t w)
{
    const uint32_t c = 0xffffffffu ^ w;
    uint32_t r = 0xffffffffu;               /* the final complement of crc32(w) */
#pragma unroll
    for (int k = 0; k < 8; k++) r ^= S->adv[k][(c >> (4 * k)) & 15] ^ S->mul[k][(acc >> (4 * k)) & 15];
    return r;
}

/* CRC register after one more byte (the <= 3 bytes of a chunk no dword covers) */
QZ_DEV uint32_t qzk_crc_byte(uint32_t c, uint32_t byte)
{
    c ^= byte;
    for (int i = 0; i < 8; i++) c = (c & 1) ? QZK_POLY ^ (c >> 1) : c >> 1;
    return c;
}

#endif
