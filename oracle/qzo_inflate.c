/*
 * qzo_inflate.c — RFC 1951 raw inflate for the oracle.  TEST INFRASTRUCTURE.
 * Stands in for zlib inflate(Z_SYNC_FLUSH) at src/qatzip_sw.c:339 (one member
 * at a time, wrapper handled by qzo_swpath.c).
 */
#include "qzo.h"
#include <string.h>

typedef struct {
    const uint8_t *in; size_t n, pos; uint64_t bb; int bc;
} br_t;

static void refill(br_t *b)
{
    while (b->bc <= 56 && b->pos < b->n) { b->bb |= (uint64_t)b->in[b->pos++] << b->bc; b->bc += 8; }
}

/* canonical Huffman decoder: 10-bit root table + per-length ranges for longer codes */
#define ROOT 10
typedef struct {
    uint16_t root[1 << ROOT];        /* (sym<<4)|len, len==0 => long code */
    uint16_t count[16], first[16], index[16];  /* canonical ranges (MSB-first codes) */
    uint16_t sorted[288];
    int maxlen;
} huff_t;

static unsigned rev(unsigned c, int len)
{
    unsigned r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (c & 1); c >>= 1; }
    return r;
}

/* returns 0 ok, -1 over-subscribed, 1 incomplete */
static int huff_build(huff_t *h, const uint8_t *lens, int n)
{
    int left = 1, used = 0; uint16_t offs[16]; unsigned code = 0;
    memset(h->count, 0, sizeof(h->count));
    memset(h->root, 0, sizeof(h->root));
    for (int i = 0; i < n; i++) h->count[lens[i]]++;
    h->count[0] = 0; h->maxlen = 0;
    for (int l = 1; l <= 15; l++) {
        left <<= 1; left -= h->count[l];
        if (left < 0) return -1;
        if (h->count[l]) h->maxlen = l;
        used += h->count[l];
    }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
    for (int i = 0; i < n; i++) if (lens[i]) h->sorted[offs[lens[i]]++] = (uint16_t)i;
    for (int l = 1, idx = 0; l <= 15; l++) {
        h->first[l] = (uint16_t)code; h->index[l] = (uint16_t)idx;
        for (int k = 0; k < h->count[l]; k++, idx++) {
            if (l <= ROOT) {
                unsigned r = rev(code + (unsigned)k, l);
                for (unsigned f = r; f < (1u << ROOT); f += 1u << l)
                    h->root[f] = (uint16_t)((h->sorted[idx] << 4) | l);
            }
        }
        code = (code + h->count[l]) << 1;
    }
    (void)used;
    return left > 0 ? 1 : 0;
}

static int huff_decode(const huff_t *h, br_t *b)
{
    unsigned e = h->root[b->bb & ((1u << ROOT) - 1)];
    if (e & 15) {
        if ((int)(e & 15) > b->bc) return -1;
        b->bb >>= (e & 15); b->bc -= (e & 15);
        return (int)(e >> 4);
    }
    /* long code: walk bit by bit, MSB-first canonical */
    unsigned code = 0; uint64_t bits = b->bb;
    for (int l = 1; l <= h->maxlen; l++) {
        code = (code << 1) | (unsigned)(bits & 1); bits >>= 1;
        if (l > b->bc) return -1;
        if (h->count[l] && code >= h->first[l] && code - h->first[l] < h->count[l]) {
            b->bb >>= l; b->bc -= l;
            return h->sorted[h->index[l] + code - h->first[l]];
        }
    }
    return -1;
}

static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t clorder[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

#define NEED(k) do { refill(&b); if (b.bc < (k)) return -1; } while (0)
#define GET(k) (tmp = (unsigned)(b.bb & ((1ull << (k)) - 1)), b.bb >>= (k), b.bc -= (k), tmp)

int qzo_inflate_raw(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *in_used, size_t *out_used)
{
    br_t b = {src, n, 0, 0, 0};
    size_t op = 0; int last; unsigned tmp;
    static __thread huff_t hl, hd;  /* per thread: bench.py times the oracle on every host core at once */
    uint8_t lens[320];

    *in_used = 0; *out_used = 0;
    do {
        unsigned type;
        NEED(3);
        last = (int)GET(1); type = GET(2);
        if (type == 0) {
            unsigned len, nlen;
            GET(b.bc & 7);
            NEED(32);
            len = GET(16); nlen = GET(16);
            if ((len ^ 0xffff) != nlen) return -1;
            /* bytes: first from the bit buffer, then straight from input */
            while (len && b.bc >= 8) { if (op >= cap) return 1; dst[op++] = (uint8_t)GET(8); len--; }
            if (len) {
                if (b.pos + len > b.n) return -1;
                if (op + len > cap) return 1;
                memcpy(dst + op, b.in + b.pos, len); op += len; b.pos += len;
            }
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            huff_build(&hl, lens, 288);
            for (i = 0; i < 30; i++) lens[i] = 5;
            huff_build(&hd, lens, 30);
        } else {
            unsigned nlen, ndist, ncode, i; uint8_t cl[19]; huff_t hc; int r;
            NEED(14);
            nlen = GET(5) + 257; ndist = GET(5) + 1; ncode = GET(4) + 4;
            if (nlen > 286 || ndist > 30) return -1;
            memset(cl, 0, sizeof(cl));
            for (i = 0; i < ncode; i++) { NEED(3); cl[clorder[i]] = (uint8_t)GET(3); }
            if (huff_build(&hc, cl, 19) != 0) return -1;
            for (i = 0; i < nlen + ndist;) {
                int sym; unsigned rep, val;
                refill(&b);
                sym = huff_decode(&hc, &b);
                if (sym < 0) return -1;
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                if (sym == 16) { if (i == 0) return -1; val = lens[i - 1]; NEED(2); rep = 3 + GET(2); }
                else if (sym == 17) { val = 0; NEED(3); rep = 3 + GET(3); }
                else { val = 0; NEED(7); rep = 11 + GET(7); }
                if (i + rep > nlen + ndist) return -1;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) return -1;
            r = huff_build(&hl, lens, (int)nlen);
            if (r < 0 || (r > 0 && hl.maxlen != 1)) return -1;
            r = huff_build(&hd, lens + nlen, (int)ndist);
            if (r < 0 || (r > 0 && hd.maxlen > 1)) return -1;
        }
        for (;;) {
            int sym;
            refill(&b);
            sym = huff_decode(&hl, &b);
            if (sym < 0) return -1;
            if (sym < 256) { if (op >= cap) return 1; dst[op++] = (uint8_t)sym; continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return -1;
            {
                unsigned len = lbase[sym], dist; int ds;
                if (lext[sym]) { if (b.bc < lext[sym]) return -1; len += GET(lext[sym]); }
                refill(&b);
                ds = huff_decode(&hd, &b);
                if (ds < 0 || ds >= 30) return -1;
                dist = dbase[ds];
                if (dext[ds]) { if (b.bc < dext[ds]) return -1; dist += GET(dext[ds]); }
                if (dist > op) return -1;
                if (op + len > cap) return 1;
                for (unsigned k = 0; k < len; k++) dst[op + k] = dst[op + k - dist];
                op += len;
            }
        }
    } while (!last);
    /* give back whole unused bytes held in the bit buffer */
    *in_used = b.pos - (size_t)(b.bc >> 3);
    *out_used = op;
    return 0;
}
