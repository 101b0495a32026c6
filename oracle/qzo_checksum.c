/*
 * qzo_checksum.c — CRC32 / crc32_combine / Adler32 / XXH32 for the oracle.
 * TEST INFRASTRUCTURE (see qzo.h).
 *
 * Reference call sites:  crc32()          src/qatzip_sw.c:219, src/qatzip.c:1711
 *                        crc32_combine()  src/qatzip_sw.c:227
 *                        XXH32()          src/qatzip_lz4.c:130 (bundled xxhash, seed 0)
 *                        adler32          zlib-internal for DEFLATE_ZLIB trailers
 * Algorithms: CRC-32/ISO-HDLC (poly 0xEDB88320 reflected, init/xorout ~0),
 * RFC 1950 Adler-32, xxHash32 (Y. Collet's public spec).
 */
#include "qzo.h"
#include <string.h>

static uint32_t crc_tab[256];
static int crc_ready;

static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_tab[i] = c;
    }
    crc_ready = 1;
}

uint32_t qzo_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    if (!crc_ready) crc_init();
    if (!p) return 0;
    crc = ~crc;
    while (n--) crc = crc_tab[(crc ^ *p++) & 0xff] ^ (crc >> 8);
    return ~crc;
}

/* GF(2) 32x32 matrix helpers: mat[i] is the image of bit i */
static uint32_t gf2_times(const uint32_t *mat, uint32_t vec)
{
    uint32_t s = 0;
    for (int i = 0; vec; vec >>= 1, i++)
        if (vec & 1) s ^= mat[i];
    return s;
}
static void gf2_square(uint32_t *sq, const uint32_t *mat)
{
    for (int i = 0; i < 32; i++) sq[i] = gf2_times(mat, mat[i]);
}

/* crc of A||B from crc(A), crc(B), len(B): advance crc(A) through len2 zero
 * bytes (operator = x^(8*len2) mod P, by repeated squaring) and xor crc(B). */
uint32_t qzo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    uint32_t even[32], odd[32];
    if (len2 == 0) return crc1;
    odd[0] = 0xEDB88320u;               /* one zero BIT */
    for (int i = 1; i < 32; i++) odd[i] = 1u << (i - 1);
    gf2_square(even, odd);              /* 2 bits */
    gf2_square(odd, even);              /* 4 bits */
    do {
        gf2_square(even, odd);          /* first pass: 8 bits = 1 byte */
        if (len2 & 1) crc1 = gf2_times(even, crc1);
        len2 >>= 1;
        if (!len2) break;
        gf2_square(odd, even);
        if (len2 & 1) crc1 = gf2_times(odd, crc1);
        len2 >>= 1;
    } while (len2);
    return crc1 ^ crc2;
}

uint32_t qzo_adler32(uint32_t adler, const uint8_t *p, size_t n)
{
    uint32_t a = adler & 0xffff, b = adler >> 16;
    if (!p) return 1;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k--) { a += *p++; b += a; }
        a %= 65521u; b %= 65521u;
    }
    return (b << 16) | a;
}

#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4 668265263u
#define P5 374761393u
static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32(const uint8_t *p)
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}

uint32_t qzo_xxh32(const uint8_t *p, size_t n, uint32_t seed)
{
    const uint8_t *end = p + n;
    uint32_t h;
    if (n >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t *lim = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1; p += 4;
            v2 = rotl(v2 + rd32(p) * P2, 13) * P1; p += 4;
            v3 = rotl(v3 + rd32(p) * P2, 13) * P1; p += 4;
            v4 = rotl(v4 + rd32(p) * P2, 13) * P1; p += 4;
        } while (p <= lim);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)n;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (*p++) * P5, 11) * P1; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
