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

/* what a K1 workgroup keeps in LDS for the CRC of its waves' chunks: the four byte tables of CRC-32 over a dword, the
 * four byte tables of the multiplication by x^(8*256) (a lane's consecutive dwords are 256 bytes apart), x^(2^i) */
typedef struct { uint32_t tab[4][256]; uint32_t ktab[4][256]; uint32_t x2n[32]; } qzk_k1crc_lds;

/* filled by the whole workgroup (any size >= 64), ends with a barrier */
QZ_DEV void qzk_k1crc_init(qzk_k1crc_lds *S)
{
    const uint32_t t0 = threadIdx.x, nt = blockDim.x;
    for (uint32_t t = t0; t < 256; t += nt) {
        uint32_t c = t;
        for (int k = 0; k < 8; k++) c = (c & 1) ? QZK_POLY ^ (c >> 1) : c >> 1;
        S->tab[0][t] = c;
    }
    if (t0 == 0) {
        uint32_t p = 1u << 30;
        S->x2n[0] = p;
        for (int i = 1; i < 32; i++) S->x2n[i] = p = qzk_multmodp(p, p);
    }
    qz_block_sync();
    const uint32_t K = qzk_x2nmodp(S->x2n, 256, 3);
    for (uint32_t t = t0; t < 256; t += nt) {
        const uint32_t c0 = S->tab[0][t];
        const uint32_t c1 = (c0 >> 8) ^ S->tab[0][c0 & 0xff];
        const uint32_t c2 = (c1 >> 8) ^ S->tab[0][c1 & 0xff];
        const uint32_t c3 = (c2 >> 8) ^ S->tab[0][c2 & 0xff];
        S->tab[1][t] = c1; S->tab[2][t] = c2; S->tab[3][t] = c3;
        for (int k = 0; k < 4; k++) S->ktab[k][t] = qzk_multmodp(K, t << (8 * k));
    }
    qz_block_sync();
}

/* acc * x^(8*256) + crc32(the four bytes of w): one Horner step of a lane over its column of the input */
QZ_DEV uint32_t qzk_k1crc_step(const qzk_k1crc_lds *S, uint32_t acc, uint32_t w)
{
    uint32_t c = 0xffffffffu ^ w;
    c = S->tab[3][c & 0xff] ^ S->tab[2][(c >> 8) & 0xff] ^ S->tab[1][(c >> 16) & 0xff] ^ S->tab[0][c >> 24];
    return S->ktab[0][acc & 0xff] ^ S->ktab[1][(acc >> 8) & 0xff] ^ S->ktab[2][(acc >> 16) & 0xff] ^ S->ktab[3][acc >> 24] ^ ~c;
}

#endif
