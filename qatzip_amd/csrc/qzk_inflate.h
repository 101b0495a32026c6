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
 * headers, marker scan) and validates the chain afterwards (qzd_inflate.hip).
 *
 * One wave per segment, many waves per CU (6 KiB LDS each).  Huffman decoding is
 * bit-serial, so the design goal is that the serial walk runs on the SCALAR unit:
 * every piece of wave-uniform state (bit buffer, positions, table entries) is pinned
 * into SGPRs with v_readfirstlane at its source, which turns the per-symbol loop into
 * s_* arithmetic and s_cbranch instead of exec-masked vector code.  The compressed
 * stream is staged 256 bytes at a time in a VGPR (one coalesced load, next window in
 * flight) and pulled with v_readlane; table construction (ballot ranking) and match
 * copies use all 64 lanes; a match's store is deferred past the next symbol's decode
 * so its L2 round trip overlaps.  Decode tables: 11-bit root (literal/length) and
 * 9-bit root (distance), u16 entries sym<<4|len, canonical walk for longer codes.
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
    uint32_t lrange[16], drange[16];       /* per code length: count<<16 | first code */
    uint16_t lindex[16], dindex[16];
    uint8_t lens[384];
    uint16_t clroot[128];
} qzk_inf_lds;

QZ_DEV uint32_t qzk_rev(uint32_t c, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (c & 1); c >>= 1; }
    return r;
}

/* Build root table + canonical ranges for `n` (<= 320) code lengths in LDS; all lanes cooperate and every
 * result used for control flow is wave-uniform.  Returns 0 ok, 1 incomplete, -1 over-subscribed. */
QZ_DEV int qzk_build_decode(const uint8_t *lens, int n, uint16_t *root, int rootbits, uint16_t *sorted,
                            uint32_t *range, uint16_t *index, int *maxlen_out, int lane)
{
    /* my symbols: lane owns i = lane + 64*k */
    uint32_t my[5];
    for (int k = 0; k < 5; k++) { int i = lane + 64 * k; my[k] = i < n ? lens[i] : 0; }
    int left = 1, maxlen = 0;
    uint32_t code = 0, offs = 0;
    for (int i = lane; i < (1 << rootbits); i += 64) root[i] = 0;
    qz_lds_sync();
    for (int L = 1; L <= 15; L++) {
        /* rank the symbols of length L in index order with ballots */
        uint32_t cntL = 0;
        for (int k = 0; k < 5; k++) {
            if (64 * k >= n) break;
            const uint64_t m = qz_ballot(my[k] == (uint32_t)L);
            if (my[k] == (uint32_t)L) {
                const uint32_t rank = cntL + (uint32_t)qz_popc64(m & qz_below(lane));
                const int sym = lane + 64 * k;
                sorted[offs + rank] = (uint16_t)sym;
                if (L <= rootbits) {
                    const uint32_t r = qzk_rev(code + rank, L);
                    for (uint32_t f = r; f < (1u << rootbits); f += 1u << L) root[f] = (uint16_t)((sym << 4) | L);
                }
            }
            cntL += (uint32_t)qz_popc64(m);
        }
        if (lane == 0) { range[L] = (cntL << 16) | (code & 0xffff); index[L] = (uint16_t)offs; }
        left <<= 1; left -= (int)cntL;
        if (left < 0) return -1;
        if (cntL) maxlen = L;
        offs += cntL;
        code = (code + cntL) << 1;
    }
    qz_lds_sync();
    *maxlen_out = maxlen;
    return left > 0 ? 1 : 0;
}

/* wave-uniform bit reader living in SGPRs; input staged in VGPR windows */
typedef struct { const uint8_t *p; uint32_t pos, end; uint64_t bb; int bc; uint32_t wbase, wcur, wnext; } qzk_bits;

QZ_DEV uint32_t qzk_winload(const qzk_bits *b, uint32_t base, int lane)
{
    uint32_t o = base + 4u * (uint32_t)lane;
    return o < b->end ? qz_ld32(b->p + o) : 0;      /* may touch <= 3 bytes past `end`: buffers carry slack */
}
QZ_DEV void qzk_bits_init(qzk_bits *b, const uint8_t *p, uint32_t pos, uint32_t end, int lane)
{
    b->p = p; b->pos = pos; b->end = end; b->bb = 0; b->bc = 0; b->wbase = pos;
    b->wcur = qzk_winload(b, pos, lane);
    b->wnext = qzk_winload(b, pos + 256, lane);
}
QZ_DEV void qzk_refill(qzk_bits *b, int lane)
{
    if (b->bc <= 32 && b->pos < b->end) {
        uint32_t idx = (b->pos - b->wbase) >> 2;
        if (idx >= 64) { b->wcur = b->wnext; b->wbase += 256; b->wnext = qzk_winload(b, b->wbase + 256, lane); idx = 0; }
        uint32_t v = qz_readlane(b->wcur, (int)idx);
        uint32_t nb = b->end - b->pos;
        if (nb > 4) nb = 4;
        if (nb < 4) v &= (1u << (8 * nb)) - 1;
        b->bb |= (uint64_t)v << b->bc; b->bc += (int)(8 * nb); b->pos += nb;
    }
}
#define QZK_GETBITS(b, k) ((uint32_t)((b)->bb & ((1ull << (k)) - 1)))
#define QZK_DROP(b, k) do { (b)->bb >>= (k); (b)->bc -= (int)(k); } while (0)

/* decode one symbol (uniform); returns -1 on invalid code / missing bits */
QZ_DEV int qzk_decode_sym(qzk_bits *b, const uint16_t *root, int rootbits, const uint16_t *sorted,
                          const uint32_t *range, const uint16_t *index, int maxlen)
{
    const uint32_t e = qz_uniform(root[(uint32_t)b->bb & ((1u << rootbits) - 1)]);
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
        const uint32_t rg = qz_uniform(range[l]), c = rg >> 16, f = rg & 0xffff;
        if (c && code >= f && code - f < c) {
            QZK_DROP(b, l);
            return (int)qz_uniform(sorted[qz_uniform(index[l]) + code - f]);
        }
    }
    return -1;
}

QZ_DEV void qzk_inflate_segment(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t sidx,
                                qzk_inf_lds *S)
{
    const int lane = qz_lane();
    /* segment record -> SGPRs */
    const qzk_infseg *sp = segs + sidx;
    const uint64_t in_off = (uint64_t)qz_uniform((uint32_t)sp->in_off) | (uint64_t)qz_uniform((uint32_t)(sp->in_off >> 32)) << 32;
    const uint64_t out_off = (uint64_t)qz_uniform((uint32_t)sp->out_off) | (uint64_t)qz_uniform((uint32_t)(sp->out_off >> 32)) << 32;
    const uint32_t in_len = qz_uniform(sp->in_len), out_cap = qz_uniform(sp->out_cap), flags = qz_uniform(sp->flags);
    const bool count_only = flags & QZK_INF_COUNT_ONLY, through = flags & QZK_INF_THROUGH_FLUSH;
    uint8_t *o = out + out_off;
    qzk_bits b; qzk_bits_init(&b, comp + in_off, 0, in_len, lane);
    uint32_t op = 0, nblocks = 0;
    int status = QZK_INF_EDATA;

    for (;;) {
        qzk_refill(&b, lane);
        if (b.bc < 3) { status = QZK_INF_EIN; break; }
        const uint32_t last = QZK_GETBITS(&b, 1); QZK_DROP(&b, 1);
        const uint32_t type = QZK_GETBITS(&b, 2); QZK_DROP(&b, 2);
        nblocks++;
        if (type == 0) {
            QZK_DROP(&b, b.bc & 7);
            qzk_refill(&b, lane);
            if (b.bc < 32) { status = QZK_INF_EIN; break; }
            const uint32_t len = QZK_GETBITS(&b, 16); QZK_DROP(&b, 16);
            const uint32_t nlen = QZK_GETBITS(&b, 16); QZK_DROP(&b, 16);
            if ((len ^ 0xffff) != nlen) { status = QZK_INF_EDATA; break; }
            /* rewind the byte position to the first unread byte, copy straight from the input */
            const uint32_t ipos = b.pos - (uint32_t)(b.bc >> 3);
            if (ipos + len > b.end) { status = QZK_INF_EIN; break; }
            if (op + len > out_cap) { status = QZK_INF_EOUT; break; }
            if (!count_only) for (uint32_t i = (uint32_t)lane; i < len; i += 64) o[op + i] = b.p[ipos + i];
            op += len;
            qzk_bits_init(&b, b.p, ipos + len, b.end, lane);        /* restart the staged window after the raw bytes */
            if (last) { status = QZK_INF_FINAL; break; }
            if (len == 0 && !through) { status = QZK_INF_FLUSH; break; }
            continue;
        }
        if (type == 3) { status = QZK_INF_EDATA; break; }
        int lmax = 0, dmax = 0;
        if (type == 1) {
            for (int i = lane; i < 288; i += 64) S->lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            if (lane < 30) S->lens[288 + lane] = 5;
            qz_lds_sync();
            qzk_build_decode(S->lens, 288, S->lroot, QZK_LROOT, S->lsorted, S->lrange, S->lindex, &lmax, lane);
            qzk_build_decode(S->lens + 288, 30, S->droot, QZK_DROOT, S->dsorted, S->drange, S->dindex, &dmax, lane);
        } else {
            qzk_refill(&b, lane);
            if (b.bc < 14) { status = QZK_INF_EIN; break; }
            const uint32_t nlen = QZK_GETBITS(&b, 5) + 257; QZK_DROP(&b, 5);
            const uint32_t ndist = QZK_GETBITS(&b, 5) + 1; QZK_DROP(&b, 5);
            const uint32_t ncode = QZK_GETBITS(&b, 4) + 4; QZK_DROP(&b, 4);
            if (nlen > 286 || ndist > 30) { status = QZK_INF_EDATA; break; }
            /* 19 code-length code lengths, 3 bits each, in the permuted order (kept in a packed constant) */
            if (lane < 19) S->lens[lane] = 0;
            qz_lds_sync();
            bool bad = false;
            for (uint32_t i = 0; i < ncode; i++) {
                qzk_refill(&b, lane);
                if (b.bc < 3) { bad = true; break; }
                const uint32_t v = QZK_GETBITS(&b, 3); QZK_DROP(&b, 3);
                /* clorder = 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 packed 5 bits each */
                const uint32_t ord = i < 6 ? ((16u | 17u << 5 | 18u << 10 | 0u << 15 | 8u << 20 | 7u << 25) >> (5 * i)) & 31
                                   : i < 12 ? ((9u | 6u << 5 | 10u << 10 | 5u << 15 | 11u << 20 | 4u << 25) >> (5 * (i - 6))) & 31
                                   : i < 18 ? ((12u | 3u << 5 | 13u << 10 | 2u << 15 | 14u << 20 | 1u << 25) >> (5 * (i - 12))) & 31 : 15u;
                if (lane == 0) S->lens[ord] = (uint8_t)v;
            }
            if (bad) { status = QZK_INF_EIN; break; }
            qz_lds_sync();
            int clmax = 0;
            if (qzk_build_decode(S->lens, 19, S->clroot, 7, S->dsorted, S->drange, S->dindex, &clmax, lane) != 0) { status = QZK_INF_EDATA; break; }
            /* run-length decode of the nlen+ndist code lengths (uniform; stores spread over lanes) */
            uint32_t i = 0, prev = 0; bad = false;
            uint8_t *Lp = S->lens + 32;      /* clear of the 19 cl lengths while decoding */
            while (i < nlen + ndist) {
                qzk_refill(&b, lane);
                const int sym = qzk_decode_sym(&b, S->clroot, 7, S->dsorted, S->drange, S->dindex, clmax);
                if (sym < 0) { bad = true; break; }
                if (sym < 16) { if (lane == 0) Lp[i] = (uint8_t)sym; prev = (uint32_t)sym; i++; continue; }
                uint32_t rep, val;
                if (sym == 16) { if (i == 0 || b.bc < 2) { bad = true; break; } val = prev; rep = 3 + QZK_GETBITS(&b, 2); QZK_DROP(&b, 2); }
                else if (sym == 17) { if (b.bc < 3) { bad = true; break; } val = 0; rep = 3 + QZK_GETBITS(&b, 3); QZK_DROP(&b, 3); }
                else { if (b.bc < 7) { bad = true; break; } val = 0; rep = 11 + QZK_GETBITS(&b, 7); QZK_DROP(&b, 7); }
                if (i + rep > nlen + ndist) { bad = true; break; }
                for (uint32_t k = (uint32_t)lane; k < rep; k += 64) Lp[i + k] = (uint8_t)val;
                prev = val; i += rep;
            }
            if (bad) { status = QZK_INF_EDATA; break; }
            qz_lds_sync();
            if (qz_uniform(Lp[256]) == 0) { status = QZK_INF_EDATA; break; }
            /* move into place: lit/len lengths at lens[0..nlen), dist lengths at lens[288..) */
            const uint8_t dl = lane < 30 ? ((uint32_t)lane < ndist ? Lp[nlen + lane] : 0) : 0;
            uint8_t ll[5];
            for (int k = 0; k < 5; k++) { int idx = lane + 64 * k; ll[k] = idx < 288 ? ((uint32_t)idx < nlen ? Lp[idx] : 0) : 0; }
            qz_lds_sync();
            for (int k = 0; k < 5; k++) { int idx = lane + 64 * k; if (idx < 288) S->lens[idx] = ll[k]; }
            if (lane < 30) S->lens[288 + lane] = dl;
            qz_lds_sync();
            int r = qzk_build_decode(S->lens, (int)nlen, S->lroot, QZK_LROOT, S->lsorted, S->lrange, S->lindex, &lmax, lane);
            if (r < 0 || (r > 0 && lmax != 1)) { status = QZK_INF_EDATA; break; }
            r = qzk_build_decode(S->lens + 288, (int)ndist, S->droot, QZK_DROOT, S->dsorted, S->drange, S->dindex, &dmax, lane);
            if (r < 0 || (r > 0 && dmax > 1)) { status = QZK_INF_EDATA; break; }
        }
        /* ---- symbol loop (scalar) ---- */
        int bstat = 2;      /* 2 = running, 1 = end of block, <0 error */
        /* a match of <= 64 bytes is loaded now and stored after the NEXT symbol has been decoded, so the L2 round
         * trip of its source bytes overlaps that decode instead of stalling the wave */
        bool pend = false; uint8_t pv = 0; uint32_t pdst = 0, plen = 0;
#define QZK_FLUSH_PENDING() do { if (pend) { if ((uint32_t)lane < plen) o[pdst + (uint32_t)lane] = pv; pend = false; } } while (0)
        while (bstat == 2) {
            qzk_refill(&b, lane);
            /* Literal runs, several symbols per LDS round trip: lane l looks up the code that would start at bit l of
             * the buffer; the real symbol boundaries are then a chain of lane hops from bit 0 (a cross-lane read each,
             * no memory), and the lanes on the chain store their literal side by side.  The chain ends at the first
             * symbol that is not a root-table literal; if that one is a length code or end-of-block with its bits
             * present, its entry is already here too.  A segment of incompressible data is ~65 K literals and is what
             * sets a call's pace: this is worth ~5 symbols a trip there. */
            int sym = -2;
            {
                const uint32_t myent = S->lroot[(uint32_t)(b.bb >> lane) & ((1u << QZK_LROOT) - 1)];
                uint64_t chain = 0; uint32_t nlit = 0; int bp = 0;
                const int avail = b.bc < 63 ? b.bc : 63;                    /* one drop of all 64 bits would be a shift by 64 */
                while (bp < avail) {
                    const uint32_t ent = qz_readlane(myent, bp);
                    const int l = (int)(ent & 15);
                    if (ent == 0 || l > avail - bp) break;                  /* long code, or its bits are not here yet */
                    if ((ent >> 4) >= 256) { sym = (int)(ent >> 4); bp += l; break; }     /* length code or end of block: its entry is here already */
                    if (op + nlit >= out_cap) break;
                    chain |= 1ull << bp; nlit++; bp += l;
                }
                if (nlit) {
                    if (!count_only) {
                        const bool mine = (chain >> lane) & 1;
                        const uint32_t at = op + (uint32_t)qz_popc64(chain & qz_below(lane));
                        QZK_FLUSH_PENDING();
                        if (mine) o[at] = (uint8_t)(myent >> 4);
                    }
                    op += nlit;
                }
                if (bp) {
                    QZK_DROP(&b, bp);
                    if (sym == -2) continue;
                    qzk_refill(&b, lane);               /* the length code's extra bits */
                }
            }
            if (sym == -2) sym = qzk_decode_sym(&b, S->lroot, QZK_LROOT, S->lsorted, S->lrange, S->lindex, lmax);
            if (sym < 0) { bstat = b.pos >= b.end && b.bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; break; }
            if (sym < 256) {
                if (op >= out_cap) { bstat = QZK_INF_EOUT; break; }
                QZK_FLUSH_PENDING();
                if (!count_only && lane == 0) o[op] = (uint8_t)sym;
                op++;
                continue;
            }
            if (sym == 256) { bstat = 1; break; }
            sym -= 257;
            if (sym >= 29) { bstat = QZK_INF_EDATA; break; }
            /* RFC 1951 length / distance bases computed arithmetically (no table loads on the serial path):
             * length symbol s: 0..7 -> 3+s, 28 -> 258, else e=(s-4)>>2 extra bits, base 3+((4+(s&3))<<e) */
            uint32_t xb = (sym < 8 || sym == 28) ? 0u : (uint32_t)(sym - 4) >> 2;
            uint32_t len = sym < 8 ? 3u + (uint32_t)sym : sym == 28 ? 258u : 3u + ((4u + ((uint32_t)sym & 3)) << xb);
            if (xb) { if ((int)xb > b.bc) { bstat = QZK_INF_EIN; break; } len += QZK_GETBITS(&b, xb); QZK_DROP(&b, xb); }
            qzk_refill(&b, lane);
            const int ds = qzk_decode_sym(&b, S->droot, QZK_DROOT, S->dsorted, S->drange, S->dindex, dmax);
            if (ds < 0 || ds >= 30) { bstat = b.pos >= b.end && b.bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; break; }
            /* distance symbol d: 0..3 -> 1+d, else e=(d-2)>>1 extra bits, base 1+((2+(d&1))<<e) */
            xb = ds < 4 ? 0u : (uint32_t)(ds - 2) >> 1;
            uint32_t dist = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1)) << xb);
            if (xb) { if ((int)xb > b.bc) { bstat = QZK_INF_EIN; break; } dist += QZK_GETBITS(&b, xb); QZK_DROP(&b, xb); }
            if (dist > op) {
                if (!through || (uint64_t)dist > out_off + op) { bstat = QZK_INF_EHIST; break; }
            }
            if (op + len > out_cap) { bstat = QZK_INF_EOUT; break; }
            QZK_FLUSH_PENDING();                        /* its bytes may be this match's source */
            if (!count_only) {
                qz_lds_sync();                          /* compiler ordering only; see qz_ld8_l2 */
                if (len <= 64) {
                    const uint32_t i = (uint32_t)lane;
                    if (i < len) {
                        /* overlapping copy: byte i comes from the already complete region, i mod dist */
                        uint32_t si = dist >= len ? i : dist == 1 ? 0u : dist == 2 ? (i & 1) : dist == 4 ? (i & 3) : i % dist;
                        pv = qz_ld8_l2(o + ((int64_t)op - dist + si));
                    }
                    pend = true; pdst = op; plen = len;
                } else {
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                        uint32_t si = dist >= len ? i : dist == 1 ? 0u : dist == 2 ? (i & 1) : dist == 4 ? (i & 3) : i % dist;
                        o[op + i] = qz_ld8_l2(o + ((int64_t)op - dist + si));
                    }
                }
            }
            op += len;
        }
        QZK_FLUSH_PENDING();
#undef QZK_FLUSH_PENDING
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

QZ_KERNEL qzk_inflate_kernel(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs)
{
    QZ_LDS qzk_inf_lds LDS[QZK_INF_WAVES];
    const uint32_t wv = qz_uniform((uint32_t)(threadIdx.x >> 6));
    const uint32_t sidx = blockIdx.x * QZK_INF_WAVES + wv;
    if (sidx >= nsegs) return;
    qzk_inflate_segment(comp, out, segs, res, sidx, &LDS[wv]);
}

#endif
