/*
 * qzk_inflate.h — K3: raw inflate (RFC 1951) of independent deflate SEGMENTS, one
 * segment per wave, gfx950.
 *
 * Replaces zlib inflate(Z_SYNC_FLUSH) on the reference's software decompress
 * path (src/qatzip_sw.c:339) and the QAT decompress request of the hardware
 * path (src/qatzip.c:2191).  A segment is what one compressed hw_buff_sz chunk
 * looks like on the wire: a run of deflate blocks that starts byte-aligned with
 * an empty history and ends either with BFINAL or with the Z_FULL_FLUSH marker
 * (empty stored block, 00 00 FF FF).  The host finds segment starts (member
 * headers, marker scan) and validates the chain afterwards (qzd_device.hip).
 *
 * One wave per segment, many waves per CU (6 KiB LDS each): the bit-serial
 * Huffman walk runs wave-uniform (SALU + broadcast LDS reads), table construction
 * and match copies are spread over the 64 lanes.  Decode tables: 11-bit root for
 * literal/length and 9-bit root for distance codes (u16 entries sym<<4|len) with a
 * canonical first-code walk for the rare longer codes.
 * CPU restatement: oracle/qzo_inflate.c.
 */
#ifndef QZK_INFLATE_H
#define QZK_INFLATE_H
#include "qzk_common.h"

#define QZK_INF_WAVES 4            /* waves (segments) per workgroup */
#define QZK_LROOT 11
#define QZK_DROOT 9

#define QZK_INF_FINAL 0            /* ended with a BFINAL block */
#define QZK_INF_FLUSH 1            /* ended at an empty stored block (flush marker) */
#define QZK_INF_EDATA (-1)         /* invalid deflate data */
#define QZK_INF_EOUT (-2)          /* output capacity exhausted */
#define QZK_INF_EIN (-3)           /* input exhausted (truncated) */
#define QZK_INF_EHIST (-4)         /* back-reference before the segment start */

#define QZK_INF_COUNT_ONLY 1u      /* flags: decode for sizes only, write nothing */
#define QZK_INF_THROUGH_FLUSH 2u   /* flags: keep going through flush markers until BFINAL (serial mode) */

typedef struct { uint64_t in_off; uint64_t out_off; uint32_t in_len; uint32_t out_cap; uint32_t flags; uint32_t pad; } qzk_infseg;
typedef struct { int32_t status; uint32_t in_used; uint32_t out_len; uint32_t nblocks; } qzk_infres;

typedef struct {
    uint16_t lroot[1 << QZK_LROOT];
    uint16_t droot[1 << QZK_DROOT];
    uint16_t lsorted[288], dsorted[32];
    uint16_t lcount[16], lfirst[16], lindex[16];
    uint16_t dcount[16], dfirst[16], dindex[16];
    uint8_t lens[384];
    uint16_t clroot[128];
    uint32_t tmp[16];
} qzk_inf_lds;

QZ_DEV uint32_t qzk_rev(uint32_t c, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (c & 1); c >>= 1; }
    return r;
}

/* Build root table + canonical ranges for `n` code lengths (all lanes cooperate).
 * Returns 0 ok, 1 incomplete, -1 over-subscribed; *maxlen_out = longest code. */
QZ_DEV int qzk_build_decode(const uint8_t *lens, int n, uint16_t *root, int rootbits, uint16_t *sorted,
                            uint16_t *count, uint16_t *first, uint16_t *index, int *maxlen_out, int lane)
{
    /* counts: tiny, done redundantly by every lane from LDS (uniform) */
    uint32_t cnt[16];
    for (int l = 0; l < 16; l++) cnt[l] = 0;
    for (int i = 0; i < n; i++) cnt[lens[i]]++;
    cnt[0] = 0;
    int left = 1, maxlen = 0;
    for (int l = 1; l <= 15; l++) {
        left <<= 1; left -= (int)cnt[l];
        if (left < 0) return -1;
        if (cnt[l]) maxlen = l;
    }
    uint32_t offs[16], code = 0, fc[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + cnt[l];
    for (int l = 1; l <= 15; l++) { fc[l] = code; code = (code + cnt[l]) << 1; }
    if (lane < 16) {
        count[lane] = (uint16_t)cnt[lane];
        first[lane] = lane ? (uint16_t)fc[lane] : 0;
        index[lane] = lane ? (uint16_t)offs[lane] : 0;
    }
    for (int i = lane; i < (1 << rootbits); i += 64) root[i] = 0;
    qz_wave_sync();
    /* symbol i: rank among symbols of the same length = number of earlier symbols with that length */
    for (int i0 = 0; i0 < n; i0 += 64) {
        int i = i0 + lane;
        int l = i < n ? lens[i] : 0;
        uint32_t rank = 0;
        if (l) for (int j = 0; j < i; j++) rank += (lens[j] == l);
        if (l) {
            sorted[offs[l] + rank] = (uint16_t)i;
            if (l <= rootbits) {
                uint32_t r = qzk_rev(fc[l] + rank, l);
                for (uint32_t f = r; f < (1u << rootbits); f += 1u << l) root[f] = (uint16_t)((i << 4) | l);
            }
        }
    }
    qz_wave_sync();
    *maxlen_out = maxlen;
    return left > 0 ? 1 : 0;
}

/* wave-uniform bit reader */
typedef struct { const uint8_t *p; uint32_t pos, end; uint64_t bb; int bc; } qzk_bits;

QZ_DEV void qzk_refill(qzk_bits *b)
{
    if (b->bc <= 32) {
        uint32_t v;
        if (b->pos + 4 <= b->end) { v = qz_ld32(b->p + b->pos); b->pos += 4; b->bb |= (uint64_t)v << b->bc; b->bc += 32; }
        else while (b->bc <= 56 && b->pos < b->end) { b->bb |= (uint64_t)b->p[b->pos++] << b->bc; b->bc += 8; }
    }
}
#define QZK_GETBITS(b, k) ((uint32_t)((b)->bb & ((1ull << (k)) - 1)))
#define QZK_DROP(b, k) do { (b)->bb >>= (k); (b)->bc -= (k); } while (0)

/* decode one symbol; returns -1 on invalid code / missing bits */
QZ_DEV int qzk_decode_sym(qzk_bits *b, const uint16_t *root, int rootbits, const uint16_t *sorted,
                          const uint16_t *count, const uint16_t *first, const uint16_t *index, int maxlen)
{
    uint32_t e = root[b->bb & ((1u << rootbits) - 1)];
    if (e) {
        int l = (int)(e & 15);
        if (l > b->bc) return -1;
        QZK_DROP(b, l);
        return (int)(e >> 4);
    }
    uint32_t code = qzk_rev((uint32_t)(b->bb & ((1u << rootbits) - 1)), rootbits);
    uint64_t bits = b->bb >> rootbits;
    for (int l = rootbits + 1; l <= maxlen; l++) {
        code = (code << 1) | (uint32_t)(bits & 1); bits >>= 1;
        if (l > b->bc) return -1;
        uint32_t c = count[l], f = first[l];
        if (c && code >= f && code - f < c) { QZK_DROP(b, l); return sorted[index[l] + code - f]; }
    }
    return -1;
}

QZ_KERNEL qzk_inflate_kernel(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs)
{
    QZ_LDS qzk_inf_lds LDS[QZK_INF_WAVES];
    const int lane = qz_lane();
    const int wv = (int)(threadIdx.x >> 6);
    const uint32_t sidx = blockIdx.x * QZK_INF_WAVES + (uint32_t)wv;
    if (sidx >= nsegs) return;
    qzk_inf_lds *S = &LDS[wv];
    const qzk_infseg sg = segs[sidx];
    const bool count_only = sg.flags & QZK_INF_COUNT_ONLY, through = sg.flags & QZK_INF_THROUGH_FLUSH;
    uint8_t *o = out + sg.out_off;
    qzk_bits b; b.p = comp + sg.in_off; b.pos = 0; b.end = sg.in_len; b.bb = 0; b.bc = 0;
    uint32_t op = 0, nblocks = 0;
    int status = QZK_INF_EDATA;
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    for (;;) {
        qzk_refill(&b);
        if (b.bc < 3) { status = QZK_INF_EIN; break; }
        const uint32_t last = QZK_GETBITS(&b, 1); QZK_DROP(&b, 1);
        const uint32_t type = QZK_GETBITS(&b, 2); QZK_DROP(&b, 2);
        nblocks++;
        if (type == 0) {
            QZK_DROP(&b, b.bc & 7);
            qzk_refill(&b);
            if (b.bc < 32) { status = QZK_INF_EIN; break; }
            uint32_t len = QZK_GETBITS(&b, 16); QZK_DROP(&b, 16);
            uint32_t nlen = QZK_GETBITS(&b, 16); QZK_DROP(&b, 16);
            if ((len ^ 0xffff) != nlen) { status = QZK_INF_EDATA; break; }
            /* rewind the byte position to the first unread byte, copy straight from the input */
            uint32_t ipos = b.pos - (uint32_t)(b.bc >> 3);
            b.bb = 0; b.bc = 0;
            if (ipos + len > b.end) { status = QZK_INF_EIN; break; }
            if (op + len > sg.out_cap) { status = QZK_INF_EOUT; break; }
            if (!count_only) for (uint32_t i = (uint32_t)lane; i < len; i += 64) o[op + i] = b.p[ipos + i];
            op += len; b.pos = ipos + len;
            if (last) { status = QZK_INF_FINAL; break; }
            if (len == 0 && !through) { status = QZK_INF_FLUSH; break; }
            continue;
        }
        if (type == 3) { status = QZK_INF_EDATA; break; }
        int lmax = 0, dmax = 0;
        if (type == 1) {
            for (int i = lane; i < 288; i += 64) S->lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            if (lane < 30) S->lens[288 + lane] = 5;
            qz_wave_sync();
            qzk_build_decode(S->lens, 288, S->lroot, QZK_LROOT, S->lsorted, S->lcount, S->lfirst, S->lindex, &lmax, lane);
            qzk_build_decode(S->lens + 288, 30, S->droot, QZK_DROOT, S->dsorted, S->dcount, S->dfirst, S->dindex, &dmax, lane);
        } else {
            qzk_refill(&b);
            if (b.bc < 14) { status = QZK_INF_EIN; break; }
            uint32_t nlen = QZK_GETBITS(&b, 5) + 257; QZK_DROP(&b, 5);
            uint32_t ndist = QZK_GETBITS(&b, 5) + 1; QZK_DROP(&b, 5);
            uint32_t ncode = QZK_GETBITS(&b, 4) + 4; QZK_DROP(&b, 4);
            if (nlen > 286 || ndist > 30) { status = QZK_INF_EDATA; break; }
            /* code-length code: 19 x 3 bits (uniform), table built by lane 0 semantics via helper */
            if (lane < 19) S->lens[lane] = 0;
            qz_wave_sync();
            bool bad = false;
            for (uint32_t i = 0; i < ncode; i++) {
                qzk_refill(&b);
                if (b.bc < 3) { bad = true; break; }
                uint32_t v = QZK_GETBITS(&b, 3); QZK_DROP(&b, 3);
                if (lane == 0) S->lens[clorder[i]] = (uint8_t)v;
            }
            if (bad) { status = QZK_INF_EIN; break; }
            qz_wave_sync();
            int clmax = 0;
            /* reuse droot's canonical arrays for the 7-bit code-length code */
            if (qzk_build_decode(S->lens, 19, S->clroot, 7, S->dsorted, S->dcount, S->dfirst, S->dindex, &clmax, lane) != 0) {
                status = QZK_INF_EDATA; break;
            }
            /* run-length decode of the nlen+ndist code lengths (uniform, lane 0 stores) */
            uint32_t i = 0; int prev = 0; bad = false;
            uint8_t *L = S->lens + 32;      /* keep clear of the 19 cl lengths while decoding */
            while (i < nlen + ndist) {
                qzk_refill(&b);
                int sym = qzk_decode_sym(&b, S->clroot, 7, S->dsorted, S->dcount, S->dfirst, S->dindex, clmax);
                if (sym < 0) { bad = true; break; }
                if (sym < 16) { if (lane == 0) L[i] = (uint8_t)sym; prev = sym; i++; continue; }
                uint32_t rep; int val;
                if (sym == 16) { if (i == 0 || b.bc < 2) { bad = true; break; } val = prev; rep = 3 + QZK_GETBITS(&b, 2); QZK_DROP(&b, 2); }
                else if (sym == 17) { if (b.bc < 3) { bad = true; break; } val = 0; rep = 3 + QZK_GETBITS(&b, 3); QZK_DROP(&b, 3); }
                else { if (b.bc < 7) { bad = true; break; } val = 0; rep = 11 + QZK_GETBITS(&b, 7); QZK_DROP(&b, 7); }
                if (i + rep > nlen + ndist) { bad = true; break; }
                for (uint32_t k = (uint32_t)lane; k < rep; k += 64) L[i + k] = (uint8_t)val;
                prev = val; i += rep;
            }
            if (bad) { status = QZK_INF_EDATA; break; }
            qz_wave_sync();
            if (L[256] == 0) { status = QZK_INF_EDATA; break; }
            /* move into place: lit/len lengths at lens[0..nlen), dist lengths at lens[288..) */
            uint8_t dl = lane < 30 ? ((uint32_t)lane < ndist ? L[nlen + lane] : 0) : 0;
            uint8_t ll[5];
            for (int k = 0; k < 5; k++) { int idx = lane + 64 * k; ll[k] = idx < 288 ? ((uint32_t)idx < nlen ? L[idx] : 0) : 0; }
            qz_wave_sync();
            for (int k = 0; k < 5; k++) { int idx = lane + 64 * k; if (idx < 288) S->lens[idx] = ll[k]; }
            if (lane < 30) S->lens[288 + lane] = dl;
            qz_wave_sync();
            int r = qzk_build_decode(S->lens, (int)nlen, S->lroot, QZK_LROOT, S->lsorted, S->lcount, S->lfirst, S->lindex, &lmax, lane);
            if (r < 0 || (r > 0 && lmax != 1)) { status = QZK_INF_EDATA; break; }
            r = qzk_build_decode(S->lens + 288, (int)ndist, S->droot, QZK_DROOT, S->dsorted, S->dcount, S->dfirst, S->dindex, &dmax, lane);
            if (r < 0 || (r > 0 && dmax > 1)) { status = QZK_INF_EDATA; break; }
        }
        /* ---- symbol loop (wave-uniform) ---- */
        int bstat = 2;      /* 2 = running, 1 = end of block, <0 error */
        while (bstat == 2) {
            qzk_refill(&b);
            int sym = qzk_decode_sym(&b, S->lroot, QZK_LROOT, S->lsorted, S->lcount, S->lfirst, S->lindex, lmax);
            if (sym < 0) { bstat = b.pos >= b.end && b.bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; break; }
            if (sym < 256) {
                if (op >= sg.out_cap) { bstat = QZK_INF_EOUT; break; }
                if (!count_only && lane == 0) o[op] = (uint8_t)sym;
                op++;
                continue;
            }
            if (sym == 256) { bstat = 1; break; }
            sym -= 257;
            if (sym >= 29) { bstat = QZK_INF_EDATA; break; }
            uint32_t len = lbase[sym], xb = lext[sym];
            if (xb) { if ((int)xb > b.bc) { bstat = QZK_INF_EIN; break; } len += QZK_GETBITS(&b, xb); QZK_DROP(&b, xb); }
            qzk_refill(&b);
            int ds = qzk_decode_sym(&b, S->droot, QZK_DROOT, S->dsorted, S->dcount, S->dfirst, S->dindex, dmax);
            if (ds < 0 || ds >= 30) { bstat = b.pos >= b.end && b.bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; break; }
            uint32_t dist = dbase[ds]; xb = dext[ds];
            if (xb) { if ((int)xb > b.bc) { bstat = QZK_INF_EIN; break; } dist += QZK_GETBITS(&b, xb); QZK_DROP(&b, xb); }
            if (dist > op) {
                if (!through || (uint64_t)dist > sg.out_off + op) { bstat = QZK_INF_EHIST; break; }
            }
            if (op + len > sg.out_cap) { bstat = QZK_INF_EOUT; break; }
            if (!count_only) {
                /* make this wave's earlier stores visible to its own loads, then copy (overlap-safe:
                 * byte i comes from the already complete region, i mod dist) */
                qz_wave_sync();
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                    uint32_t si = dist >= len ? i : i % dist;
                    o[op + i] = o[(int64_t)op - dist + si];
                }
            }
            op += len;
        }
        if (bstat != 1) { status = bstat; break; }
        if (last) { status = QZK_INF_FINAL; break; }
    }
    if (lane == 0) {
        qzk_infres r;
        r.status = status; r.out_len = op; r.nblocks = nblocks;
        r.in_used = b.pos - (uint32_t)(b.bc >> 3);     /* whole unused bytes go back */
        res[sidx] = r;
    }
}

#endif
