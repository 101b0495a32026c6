/*
 * qzo_lz4.c — LZ4 block + frame restatement for the oracle.  TEST INFRASTRUCTURE.
 *
 * Reference call sites: LZ4F_compressFrame src/qatzip_sw.c:451-456 with
 * {contentChecksumFlag=1, contentSize=src_len, autoFlush=1, level=comp_lvl};
 * LZ4F_decompress src/qatzip_sw.c:496.  lz4 is not vendored in /root/reference
 * (pinned: liblz4 1.9.3); this restates the published block/frame formats and
 * the LZ4_compress_fast (acceleration 1) parse: for inputs < 64 KB + 11 a 13-bit
 * multiplicative hash of 4 bytes into a u16 position table that starts zeroed
 * (so position 0 is a live candidate), search step growing by one every 64
 * misses, backward extension, MFLIMIT 12 / LASTLITERALS 5, and the
 * "dstCapacity = srcSize-1 else stored" rule of the frame layer.
 *
 * Frames whose content exceeds one 64 KB block use lz4's linked-block mode:
 * LZ4F_compressFrame drives LZ4_compress_fast_continue over the blocks as they lie in
 * the source (stableSrc), i.e. ONE parse state for the whole frame - a 4096-entry table
 * of 32-bit positions (frame-relative), the 5-byte multiplicative hash of the 64-bit
 * build, candidates valid up to 65535 bytes back and across block borders (prefix
 * mode) - restarted at every block: first byte inserted unsearched, no match begins
 * in a block's last 12 bytes or extends into its last 5, a block that does not shrink
 * is stored while the table keeps what its attempt inserted (qzo_lz4_linked_block).
 */
#include "qzo.h"
#include <string.h>

#define MINMATCH 4
#define MFLIMIT 12
#define LASTLITERALS 5
#define LZ4_64KLIMIT (65536 + (MFLIMIT - 1))
#define ML_BITS 4
#define ML_MASK 15u
#define RUN_MASK 15u

static uint32_t rd32(const uint8_t *p)
{
    return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
}
static void wr32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}
#define HASH16(seq) (((seq) * 2654435761u) >> 19)

static unsigned count_match(const uint8_t *in, const uint8_t *match, const uint8_t *limit)
{
    const uint8_t *s = in;
    while (in < limit && *in == *match) { in++; match++; }
    return (unsigned)(in - s);
}

int qzo_lz4_compress_block(const uint8_t *src, int n, uint8_t *dst, int cap)
{
    static __thread uint16_t table[8192];       /* per thread (bench.py runs the oracle on every host core at once) */
    const uint8_t *ip = src, *anchor = src, *const iend = src + n;
    const uint8_t *const mflimit_p1 = iend - MFLIMIT + 1, *const matchlimit = iend - LASTLITERALS;
    uint8_t *op = dst, *const olimit = dst + cap, *token;
    uint32_t forward_h;

    if (n >= LZ4_64KLIMIT || n < 0) return 0;
    memset(table, 0, sizeof(table));
    if (n < MFLIMIT + 1) goto last_literals;

    table[HASH16(rd32(ip))] = 0;
    ip++; forward_h = HASH16(rd32(ip));

    for (;;) {
        const uint8_t *match;
        {
            const uint8_t *forward_ip = ip; int step = 1, search_nb = 1 << 6;
            do {
                uint32_t h = forward_h, cur = (uint32_t)(forward_ip - src), mi = table[h];
                ip = forward_ip; forward_ip += step; step = search_nb++ >> 6;
                if (forward_ip > mflimit_p1) goto last_literals;
                match = src + mi;
                forward_h = HASH16(rd32(forward_ip));
                table[h] = (uint16_t)cur;
            } while (rd32(match) != rd32(ip));
        }
        while (ip > anchor && match > src && ip[-1] == match[-1]) { ip--; match--; }
        {
            unsigned lit = (unsigned)(ip - anchor);
            token = op++;
            if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > olimit) return 0;
            if (lit >= RUN_MASK) {
                int len = (int)(lit - RUN_MASK);
                *token = RUN_MASK << ML_BITS;
                for (; len >= 255; len -= 255) *op++ = 255;
                *op++ = (uint8_t)len;
            } else *token = (uint8_t)(lit << ML_BITS);
            memcpy(op, anchor, lit); op += lit;
        }
next_match:
        op[0] = (uint8_t)(ip - match); op[1] = (uint8_t)((ip - match) >> 8); op += 2;
        {
            unsigned mc = count_match(ip + MINMATCH, match + MINMATCH, matchlimit);
            ip += mc + MINMATCH;
            if (op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit) return 0;
            if (mc >= ML_MASK) {
                *token += ML_MASK; mc -= ML_MASK;
                while (mc >= 255) { *op++ = 255; mc -= 255; }
                *op++ = (uint8_t)mc;
            } else *token += (uint8_t)mc;
        }
        anchor = ip;
        if (ip >= mflimit_p1) break;
        table[HASH16(rd32(ip - 2))] = (uint16_t)(ip - 2 - src);
        {
            uint32_t h = HASH16(rd32(ip)), cur = (uint32_t)(ip - src), mi = table[h];
            match = src + mi;
            table[h] = (uint16_t)cur;
            if (rd32(match) == rd32(ip)) { token = op++; *token = 0; goto next_match; }
        }
        forward_h = HASH16(rd32(++ip));
    }
last_literals:
    {
        size_t last_run = (size_t)(iend - anchor);
        if (op + last_run + 1 + (last_run + 255 - RUN_MASK) / 255 > olimit) return 0;
        if (last_run >= RUN_MASK) {
            size_t acc = last_run - RUN_MASK;
            *op++ = RUN_MASK << ML_BITS;
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else *op++ = (uint8_t)(last_run << ML_BITS);
        memcpy(op, anchor, last_run); op += last_run;
    }
    return (int)(op - dst);
}

/* LZ4_decompress_safe with `prefix` bytes of history valid before dst */
static int lz4_dec(const uint8_t *src, int n, uint8_t *dst, int cap, size_t prefix)
{
    const uint8_t *ip = src, *const iend = src + n;
    uint8_t *op = dst, *const oend = dst + cap;
    if (n == 0) return -1;
    for (;;) {
        unsigned token, len, off;
        if (ip >= iend) return -1;
        token = *ip++;
        len = token >> 4;
        if (len == 15) { unsigned b; do { if (ip >= iend) return -1; b = *ip++; len += b; } while (b == 255); }
        if (len > (size_t)(iend - ip) || len > (size_t)(oend - op)) return -1;
        memcpy(op, ip, len); op += len; ip += len;
        if (ip == iend) break;                /* last sequence: literals only */
        if (iend - ip < 2) return -1;
        off = ip[0] | (unsigned)ip[1] << 8; ip += 2;
        if (off == 0 || off > (size_t)(op - dst) + prefix) return -1;
        len = token & 15;
        if (len == 15) { unsigned b; do { if (ip >= iend) return -1; b = *ip++; len += b; } while (b == 255); }
        len += MINMATCH;
        if (len > (size_t)(oend - op)) return -1;
        for (unsigned k = 0; k < len; k++) op[k] = op[(ptrdiff_t)k - (ptrdiff_t)off];
        op += len;
    }
    return (int)(op - dst);
}

int qzo_lz4_decompress_block(const uint8_t *src, int n, uint8_t *dst, int cap)
{
    return lz4_dec(src, n, dst, cap, 0);
}
int qzo_lz4_decompress_block_prefix(const uint8_t *src, int n, uint8_t *dst, int cap, size_t prefix)
{
    return lz4_dec(src, n, dst, cap, prefix);
}

/* One block [bs, bs + n) of a linked-block frame that starts at `base`; table = the frame's 4096 x u32 parse state.
 * lz4 1.9.3 LZ4_compress_generic_validated(byU32, first block: usingExtDict with an empty dictionary, later ones:
 * withPrefix64k, noDictIssue - a 64 KB block never leaves a "small" dictionary). */
static uint32_t hash5(const uint8_t *p)       /* lz4's LZ4_hash5 (64-bit little-endian builds): the low five bytes, 12 bits */
{
    const uint64_t seq = (uint64_t)rd32(p) | (uint64_t)p[4] << 32;
    return (uint32_t)(((seq << 24) * 889523592379ULL) >> 52);
}
#define HASH5(p) hash5(p)
static int qzo_lz4_linked_block(const uint8_t *base, size_t bs, int n, uint8_t *dst, int cap, uint32_t *table)
{
    const uint8_t *const source = base + bs;
    const uint8_t *ip = source, *anchor = source, *const iend = source + n;
    const uint8_t *const mflimit_p1 = iend - MFLIMIT + 1, *const matchlimit = iend - LASTLITERALS;
    uint8_t *op = dst, *const olimit = dst + cap, *token;
    uint32_t forward_h;

    if (n < MFLIMIT + 1) goto last_literals;
    table[HASH5(ip)] = (uint32_t)(ip - base);
    ip++; forward_h = HASH5(ip);
    for (;;) {
        const uint8_t *match;
        {
            const uint8_t *forward_ip = ip; int step = 1, search_nb = 1 << 6;
            for (;;) {
                uint32_t h = forward_h, cur = (uint32_t)(forward_ip - base), mi = table[h];
                ip = forward_ip; forward_ip += step; step = search_nb++ >> 6;
                if (forward_ip > mflimit_p1) goto last_literals;
                match = base + mi;
                forward_h = HASH5(forward_ip);
                table[h] = cur;
                if (mi + 65535u < cur) continue;                 /* too far */
                if (rd32(match) == rd32(ip)) break;
            }
        }
        while (ip > anchor && match > base && ip[-1] == match[-1]) { ip--; match--; }
        {
            unsigned lit = (unsigned)(ip - anchor);
            token = op++;
            if (op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > olimit) return 0;
            if (lit >= RUN_MASK) {
                int len = (int)(lit - RUN_MASK);
                *token = RUN_MASK << ML_BITS;
                for (; len >= 255; len -= 255) *op++ = 255;
                *op++ = (uint8_t)len;
            } else *token = (uint8_t)(lit << ML_BITS);
            memcpy(op, anchor, lit); op += lit;
        }
next_match:
        op[0] = (uint8_t)(ip - match); op[1] = (uint8_t)((ip - match) >> 8); op += 2;
        {
            unsigned mc = count_match(ip + MINMATCH, match + MINMATCH, matchlimit);
            ip += mc + MINMATCH;
            if (op + (1 + LASTLITERALS) + (mc + 240) / 255 > olimit) return 0;
            if (mc >= ML_MASK) {
                *token += ML_MASK; mc -= ML_MASK;
                while (mc >= 255) { *op++ = 255; mc -= 255; }
                *op++ = (uint8_t)mc;
            } else *token += (uint8_t)mc;
        }
        anchor = ip;
        if (ip >= mflimit_p1) break;
        table[HASH5(ip - 2)] = (uint32_t)(ip - 2 - base);
        {
            uint32_t h = HASH5(ip), cur = (uint32_t)(ip - base), mi = table[h];
            match = base + mi;
            table[h] = cur;
            if (mi + 65535u >= cur && rd32(match) == rd32(ip)) { token = op++; *token = 0; goto next_match; }
        }
        ip++; forward_h = HASH5(ip);
    }
last_literals:
    {
        size_t last_run = (size_t)(iend - anchor);
        if (op + last_run + 1 + (last_run + 255 - RUN_MASK) / 255 > olimit) return 0;
        if (last_run >= RUN_MASK) {
            size_t acc = last_run - RUN_MASK;
            *op++ = RUN_MASK << ML_BITS;
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else *op++ = (uint8_t)(last_run << ML_BITS);
        memcpy(op, anchor, last_run); op += last_run;
    }
    return (int)(op - dst);
}

/* LZ4F_compressFrameBound for these prefs: 19 B max header + 4 B per block + content + endmark + checksum */
size_t qzo_lz4f_bound(size_t n) { return 19 + 4 * ((n >> 16) + ((n & 65535) != 0)) + n + 8; }

size_t qzo_lz4f_compress_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
    size_t pos = 0; unsigned flg;
    if (cap < qzo_lz4f_bound(n)) return 0;          /* LZ4F_ERROR_dstMaxSize_tooSmall */
    if (n > 65536) {
        /* linked blocks (FLG 0x4C): one parse state across the blocks of the frame.  Positions are 32-bit: lz4 rescales
         * its table past 2 GiB (LZ4_renormDictT), which is not restated - such calls are not linked frames here */
        static __thread uint32_t table[4096];
        if (n > 0x7fff0000u) return 0;
        memset(table, 0, sizeof(table));
        wr32(dst, 0x184D2204u); pos = 4;
        dst[pos++] = (uint8_t)((1u << 6) | (1u << 3) | (1u << 2));      /* v1, blocks linked, content size, content checksum */
        dst[pos++] = 4u << 4;
        wr32(dst + pos, (uint32_t)n); wr32(dst + pos + 4, (uint32_t)((uint64_t)n >> 32)); pos += 8;
        dst[pos] = (uint8_t)(qzo_xxh32(dst + 4, pos - 4, 0) >> 8); pos++;
        for (size_t bs = 0; bs < n; bs += 65536) {
            const int bn = (int)(n - bs < 65536 ? n - bs : 65536);
            const int c = qzo_lz4_linked_block(src, bs, bn, dst + pos + 4, bn - 1, table);
            if (c == 0) { wr32(dst + pos, (uint32_t)bn | 0x80000000u); memcpy(dst + pos + 4, src + bs, (size_t)bn); pos += 4 + (size_t)bn; }
            else { wr32(dst + pos, (uint32_t)c); pos += 4 + (size_t)c; }
        }
        wr32(dst + pos, 0); pos += 4;
        wr32(dst + pos, qzo_xxh32(src, n, 0)); pos += 4;
        return pos;
    }
    wr32(dst, 0x184D2204u); pos = 4;
    flg = (1u << 6) | (1u << 5) | (n ? 1u << 3 : 0) | (1u << 2);   /* v1, independent, [csize], ccheck */
    dst[pos++] = (uint8_t)flg;
    dst[pos++] = 4u << 4;                          /* 64 KB blocks */
    if (n) { wr32(dst + pos, (uint32_t)n); wr32(dst + pos + 4, (uint32_t)((uint64_t)n >> 32)); pos += 8; }
    dst[pos] = (uint8_t)(qzo_xxh32(dst + 4, pos - 4, 0) >> 8); pos++;
    if (n) {
        int c = qzo_lz4_compress_block(src, (int)n, dst + pos + 4, (int)n - 1);
        if (c == 0) { wr32(dst + pos, (uint32_t)n | 0x80000000u); memcpy(dst + pos + 4, src, n); pos += 4 + n; }
        else { wr32(dst + pos, (uint32_t)c); pos += 4 + (size_t)c; }
    }
    wr32(dst + pos, 0); pos += 4;
    wr32(dst + pos, qzo_xxh32(src, n, 0)); pos += 4;
    return pos;
}
