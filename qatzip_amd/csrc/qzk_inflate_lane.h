/*
 * qzk_inflate_lane.h — K3b: raw inflate with ONE SEGMENT PER LANE, gfx950.
 *
 * Same contract as qzk_inflate_kernel (qzk_inflate.h: segment records, status codes,
 * count-only / through-flush flags) and the same place in the reference
 * (zlib inflate(), src/qatzip_sw.c:339), but the opposite mapping: Huffman decoding
 * is bit-serial, and a 2 GiB call is 32 768 independent segments, so every LANE
 * decodes its own segment (64 segments per wave, no cross-lane traffic, no LDS).
 * The wave-per-segment kernel is bound by the CU's single scalar unit; this one
 * spreads the same serial work over the vector lanes and wins as soon as a call has
 * thousands of segments (the host picks the kernel by segment count).
 *
 * Per lane: a 64-bit bit buffer refilled with 4-byte loads from its own stream,
 * decode tables in a per-segment HBM scratch record (11-bit / 9-bit root tables +
 * canonical ranges, 6.3 KiB, L2/Infinity-Cache resident), output bytes to its own
 * region.  Divergence is bounded by a small state machine: per loop trip a lane
 * either parses a block header, decodes one symbol, or copies <= 8 bytes of its
 * pending match / stored run, so a long copy in one lane does not stall the others.
 */
#ifndef QZK_INFLATE_LANE_H
#define QZK_INFLATE_LANE_H
#include "qzk_inflate.h"

typedef struct {
    uint16_t lroot[1 << QZK_LROOT];
    uint16_t droot[1 << QZK_DROOT];
    uint16_t lsorted[288], dsorted[32];
    uint16_t lcount[16], lfirst[16], lindex[16];
    uint16_t dcount[16], dfirst[16], dindex[16];
    uint8_t lens[320];
} qzk_inf_tab;

/* serial (per-lane) canonical table build; returns 0 ok, 1 incomplete, -1 over-subscribed */
QZ_DEV int qzk_lane_build(const uint8_t *lens, int n, uint16_t *root, int rootbits, uint16_t *sorted,
                          uint16_t *count, uint16_t *first, uint16_t *index, int *maxlen_out)
{
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1, maxlen = 0; uint32_t code = 0, off = 0;
    for (int l = 1; l <= 15; l++) {
        const uint32_t c = count[l];
        left <<= 1; left -= (int)c;
        if (left < 0) return -1;
        if (c) maxlen = l;
        first[l] = (uint16_t)code; index[l] = (uint16_t)off;
        code = (code + c) << 1; off += c;
    }
    for (int i = 0; i < (1 << rootbits); i += 2) *(uint32_t *)(root + i) = 0;
    uint16_t next[16];
    for (int l = 1; l <= 15; l++) next[l] = 0;
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (!l) continue;
        const uint32_t rank = next[l]++;
        sorted[index[l] + rank] = (uint16_t)i;
        if (l <= rootbits) {
            const uint32_t r = qzk_rev(first[l] + rank, l);
            for (uint32_t f = r; f < (1u << rootbits); f += 1u << l) root[f] = (uint16_t)((i << 4) | l);
        }
    }
    *maxlen_out = maxlen;
    return left > 0 ? 1 : 0;
}

typedef struct { const uint8_t *p; uint32_t pos, end; uint64_t bb; int bc; } qzk_lbits;

QZ_DEV void qzk_lrefill(qzk_lbits *b)
{
    if (b->bc <= 32) {
        if (b->pos + 4 <= b->end) { b->bb |= (uint64_t)qz_ld32(b->p + b->pos) << b->bc; b->pos += 4; b->bc += 32; }
        else while (b->bc <= 56 && b->pos < b->end) { b->bb |= (uint64_t)b->p[b->pos++] << b->bc; b->bc += 8; }
    }
}

QZ_DEV int qzk_ldecode(qzk_lbits *b, const uint16_t *root, int rootbits, const uint16_t *sorted,
                       const uint16_t *count, const uint16_t *first, const uint16_t *index, int maxlen)
{
    const uint32_t e = root[(uint32_t)b->bb & ((1u << rootbits) - 1)];
    if (e) {
        const int l = (int)(e & 15);
        if (l > b->bc) return -1;
        QZK_DROP(b, l);
        return (int)(e >> 4);
    }
    uint32_t code = qzk_rev((uint32_t)b->bb & ((1u << rootbits) - 1), rootbits);
    uint64_t bits = b->bb >> rootbits;
    for (int l = rootbits + 1; l <= maxlen; l++) {
        code = (code << 1) | (uint32_t)(bits & 1); bits >>= 1;
        if (l > b->bc) return -1;
        const uint32_t c = count[l], f = first[l];
        if (c && code >= f && code - f < c) { QZK_DROP(b, l); return sorted[index[l] + code - f]; }
    }
    return -1;
}

/* per-lane output staging: bytes are collected in a 64-bit register and leave as one 8-byte store, so a lane
 * issues one memory request per 8 output bytes instead of one per byte */
typedef struct __attribute__((packed, aligned(1))) { uint64_t v; } qz_u64u;
typedef struct { uint8_t *o; uint32_t opf, on, cap; uint64_t buf; } qzk_lout;

/* make everything appended so far visible in memory (a match is about to read it back) */
QZ_DEV void qzk_lout_sync(qzk_lout *w)
{
    if (!w->on) return;
    if (w->opf + 8 <= w->cap) ((qz_u64u *)(w->o + w->opf))->v = w->buf;      /* bytes beyond `on` get rewritten later */
    else for (uint32_t i = 0; i < w->on; i++) w->o[w->opf + i] = (uint8_t)(w->buf >> (8 * i));
}
/* append the low k (1..8) bytes of v */
QZ_DEV void qzk_lout_put(qzk_lout *w, uint64_t v, uint32_t k)
{
    if (k < 8) v &= (1ull << (8 * k)) - 1;
    w->buf |= v << (8 * w->on);
    const uint32_t tot = w->on + k;
    if (tot >= 8) {
        if (w->opf + 8 <= w->cap) ((qz_u64u *)(w->o + w->opf))->v = w->buf;
        else for (uint32_t i = 0; i < 8 && w->opf + i < w->cap; i++) w->o[w->opf + i] = (uint8_t)(w->buf >> (8 * i));
        w->buf = w->on ? v >> (8 * (8 - w->on)) : 0;
        w->opf += 8; w->on = tot - 8;
    } else w->on = tot;
}
QZ_DEV uint64_t qzk_ld64u(const uint8_t *p) { return ((const qz_u64u *)p)->v; }

enum { QZK_LS_HDR = 0, QZK_LS_SYM, QZK_LS_COPY, QZK_LS_RAW, QZK_LS_DONE };

QZ_KERNEL qzk_inflate_lane_kernel(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res,
                                  uint32_t nsegs, qzk_inf_tab *tabs)
{
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nsegs) return;
    const qzk_infseg sg = segs[sidx];
    qzk_inf_tab *T = tabs + sidx;
    const bool count_only = sg.flags & QZK_INF_COUNT_ONLY, through = sg.flags & QZK_INF_THROUGH_FLUSH;
    uint8_t *o = out + sg.out_off;
    qzk_lbits b; b.p = comp + sg.in_off; b.pos = 0; b.end = sg.in_len; b.bb = 0; b.bc = 0;
    uint32_t op = 0, nblocks = 0, last = 0;
    uint32_t clen = 0, cdist = 0, rpos = 0;             /* pending copy: length, distance (match) / input position (raw) */
    int lmax = 0, dmax = 0, status = QZK_INF_EDATA, state = QZK_LS_HDR;
    qzk_lout w; w.o = o; w.opf = 0; w.on = 0; w.cap = sg.out_cap; w.buf = 0;

    while (state != QZK_LS_DONE) {
        if (state == QZK_LS_COPY) {
            /* <= 8 bytes of the pending match; source bytes all precede op (i mod dist), so no intra-step hazard */
            const uint32_t k = clen < 8 ? clen : 8;
            if (!count_only) {
                const uint8_t *s = o + ((int64_t)op - (int64_t)cdist);     /* may reach before o in through mode */
                uint64_t v;
                if (cdist >= 8) v = qzk_ld64u(s);
                else { v = 0; for (uint32_t i = 0; i < k; i++) v |= (uint64_t)s[i % cdist] << (8 * i); }
                qzk_lout_put(&w, v, k);
                /* the next step of this copy may read what this one produced */
                if (cdist < 16) qzk_lout_sync(&w);
            }
            op += k; clen -= k;
            if (!clen) state = QZK_LS_SYM;
            continue;
        }
        if (state == QZK_LS_RAW) {
            const uint32_t k = clen < 8 ? clen : 8;
            if (!count_only) {
                uint64_t v = 0;
                if (rpos + 8 <= b.end) v = qzk_ld64u(b.p + rpos);
                else for (uint32_t i = 0; i < k; i++) v |= (uint64_t)b.p[rpos + i] << (8 * i);
                qzk_lout_put(&w, v, k);
            }
            op += k; rpos += k; clen -= k;
            if (!clen) { if (last) { status = QZK_INF_FINAL; state = QZK_LS_DONE; } else state = QZK_LS_HDR; }
            continue;
        }
        if (state == QZK_LS_HDR) {
            qzk_lrefill(&b);
            if (b.bc < 3) { status = QZK_INF_EIN; break; }
            last = QZK_GETBITS(&b, 1); QZK_DROP(&b, 1);
            const uint32_t type = QZK_GETBITS(&b, 2); QZK_DROP(&b, 2);
            nblocks++;
            if (type == 0) {
                QZK_DROP(&b, b.bc & 7);
                qzk_lrefill(&b);
                if (b.bc < 32) { status = QZK_INF_EIN; break; }
                const uint32_t len = QZK_GETBITS(&b, 16); QZK_DROP(&b, 16);
                const uint32_t nlen = QZK_GETBITS(&b, 16); QZK_DROP(&b, 16);
                if ((len ^ 0xffff) != nlen) { status = QZK_INF_EDATA; break; }
                const uint32_t ipos = b.pos - (uint32_t)(b.bc >> 3);
                b.bb = 0; b.bc = 0; b.pos = ipos + len;
                if (ipos + len > b.end) { status = QZK_INF_EIN; break; }
                if (op + len > sg.out_cap) { status = QZK_INF_EOUT; break; }
                if (len == 0) {
                    if (last) { status = QZK_INF_FINAL; break; }
                    if (!through) { status = QZK_INF_FLUSH; break; }
                    continue;
                }
                clen = len; rpos = ipos; state = QZK_LS_RAW;
                continue;
            }
            if (type == 3) { status = QZK_INF_EDATA; break; }
            if (type == 1) {
                for (int i = 0; i < 288; i++) T->lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                qzk_lane_build(T->lens, 288, T->lroot, QZK_LROOT, T->lsorted, T->lcount, T->lfirst, T->lindex, &lmax);
                for (int i = 0; i < 30; i++) T->lens[i] = 5;
                qzk_lane_build(T->lens, 30, T->droot, QZK_DROOT, T->dsorted, T->dcount, T->dfirst, T->dindex, &dmax);
            } else {
                qzk_lrefill(&b);
                if (b.bc < 14) { status = QZK_INF_EIN; break; }
                const uint32_t nlen = QZK_GETBITS(&b, 5) + 257; QZK_DROP(&b, 5);
                const uint32_t ndist = QZK_GETBITS(&b, 5) + 1; QZK_DROP(&b, 5);
                const uint32_t ncode = QZK_GETBITS(&b, 4) + 4; QZK_DROP(&b, 4);
                if (nlen > 286 || ndist > 30) { status = QZK_INF_EDATA; break; }
                for (int i = 0; i < 19; i++) T->lens[i] = 0;
                bool bad = false;
                for (uint32_t i = 0; i < ncode; i++) {
                    qzk_lrefill(&b);
                    if (b.bc < 3) { bad = true; break; }
                    const uint32_t v = QZK_GETBITS(&b, 3); QZK_DROP(&b, 3);
                    const uint32_t ord = i < 6 ? ((16u | 17u << 5 | 18u << 10 | 0u << 15 | 8u << 20 | 7u << 25) >> (5 * i)) & 31
                                       : i < 12 ? ((9u | 6u << 5 | 10u << 10 | 5u << 15 | 11u << 20 | 4u << 25) >> (5 * (i - 6))) & 31
                                       : i < 18 ? ((12u | 3u << 5 | 13u << 10 | 2u << 15 | 14u << 20 | 1u << 25) >> (5 * (i - 12))) & 31 : 15u;
                    T->lens[ord] = (uint8_t)v;
                }
                if (bad) { status = QZK_INF_EIN; break; }
                int clmax = 0;
                /* the 7-bit code-length code borrows the distance-table arrays */
                if (qzk_lane_build(T->lens, 19, T->droot, 7, T->dsorted, T->dcount, T->dfirst, T->dindex, &clmax) != 0) { status = QZK_INF_EDATA; break; }
                uint32_t i = 0, prev = 0;
                uint8_t *L = T->lens;               /* final place: [0, nlen) lit/len, [nlen, nlen+ndist) distance */
                while (i < nlen + ndist) {
                    qzk_lrefill(&b);
                    const int sym = qzk_ldecode(&b, T->droot, 7, T->dsorted, T->dcount, T->dfirst, T->dindex, clmax);
                    if (sym < 0) { bad = true; break; }
                    if (sym < 16) { L[i++] = (uint8_t)sym; prev = (uint32_t)sym; continue; }
                    uint32_t rep, val;
                    if (sym == 16) { if (i == 0 || b.bc < 2) { bad = true; break; } val = prev; rep = 3 + QZK_GETBITS(&b, 2); QZK_DROP(&b, 2); }
                    else if (sym == 17) { if (b.bc < 3) { bad = true; break; } val = 0; rep = 3 + QZK_GETBITS(&b, 3); QZK_DROP(&b, 3); }
                    else { if (b.bc < 7) { bad = true; break; } val = 0; rep = 11 + QZK_GETBITS(&b, 7); QZK_DROP(&b, 7); }
                    if (i + rep > nlen + ndist) { bad = true; break; }
                    for (uint32_t k = 0; k < rep; k++) L[i + k] = (uint8_t)val;
                    prev = val; i += rep;
                }
                if (bad) { status = QZK_INF_EDATA; break; }
                if (L[256] == 0) { status = QZK_INF_EDATA; break; }
                int r = qzk_lane_build(L, (int)nlen, T->lroot, QZK_LROOT, T->lsorted, T->lcount, T->lfirst, T->lindex, &lmax);
                if (r < 0 || (r > 0 && lmax != 1)) { status = QZK_INF_EDATA; break; }
                r = qzk_lane_build(L + nlen, (int)ndist, T->droot, QZK_DROOT, T->dsorted, T->dcount, T->dfirst, T->dindex, &dmax);
                if (r < 0 || (r > 0 && dmax > 1)) { status = QZK_INF_EDATA; break; }
            }
            state = QZK_LS_SYM;
            continue;
        }
        /* ---- QZK_LS_SYM: one symbol ---- */
        qzk_lrefill(&b);
        int sym = qzk_ldecode(&b, T->lroot, QZK_LROOT, T->lsorted, T->lcount, T->lfirst, T->lindex, lmax);
        if (sym < 0) { status = b.pos >= b.end && b.bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; break; }
        if (sym < 256) {
            if (op >= sg.out_cap) { status = QZK_INF_EOUT; break; }
            if (!count_only) qzk_lout_put(&w, (uint64_t)sym, 1);
            op++;
            continue;
        }
        if (sym == 256) {
            if (last) { status = QZK_INF_FINAL; break; }
            state = QZK_LS_HDR;
            continue;
        }
        sym -= 257;
        if (sym >= 29) { status = QZK_INF_EDATA; break; }
        uint32_t xb = (sym < 8 || sym == 28) ? 0u : (uint32_t)(sym - 4) >> 2;
        uint32_t len = sym < 8 ? 3u + (uint32_t)sym : sym == 28 ? 258u : 3u + ((4u + ((uint32_t)sym & 3)) << xb);
        if (xb) { if ((int)xb > b.bc) { status = QZK_INF_EIN; break; } len += QZK_GETBITS(&b, xb); QZK_DROP(&b, xb); }
        qzk_lrefill(&b);
        const int ds = qzk_ldecode(&b, T->droot, QZK_DROOT, T->dsorted, T->dcount, T->dfirst, T->dindex, dmax);
        if (ds < 0 || ds >= 30) { status = b.pos >= b.end && b.bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; break; }
        xb = ds < 4 ? 0u : (uint32_t)(ds - 2) >> 1;
        uint32_t dist = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1)) << xb);
        if (xb) { if ((int)xb > b.bc) { status = QZK_INF_EIN; break; } dist += QZK_GETBITS(&b, xb); QZK_DROP(&b, xb); }
        if (dist > op) {
            if (!through || (uint64_t)dist > sg.out_off + op) { status = QZK_INF_EHIST; break; }
        }
        if (op + len > sg.out_cap) { status = QZK_INF_EOUT; break; }
        clen = len; cdist = dist; state = QZK_LS_COPY;
        if (!count_only) qzk_lout_sync(&w);            /* the source of the copy may still be in the staging register */
    }
    if (!count_only) for (uint32_t i = 0; i < w.on; i++) w.o[w.opf + i] = (uint8_t)(w.buf >> (8 * i));
    qzk_infres r;
    r.status = status; r.out_len = op; r.nblocks = nblocks;
    r.in_used = b.pos - (uint32_t)(b.bc >> 3);
    res[sidx] = r;
}

#endif
