/*
 * qzk_deflate_lazy.h — zlib's lazy parser (deflate_slow, comp_lvl 4-9) taken apart into its parallel pieces, gfx950.
 *
 * Same place in the reference as K1/K1b (zlib deflate() behind qzDeflateSWCompress, src/qatzip_sw.c:147-231) and the
 * same output contract (symbol arrays + qzk_lzmeta, consumed by K2).  What makes the lazy levels different from the
 * greedy ones: deflate_slow puts EVERY position into the hash chains (the one at the loop top, and all of an emitted
 * match's), so the chains do not depend on the parse.  Hence three steps:
 *
 *   L1 qzk_lazy_chain_kernel   prev[] for the whole chunk: distance from each position to the previous one with the
 *                              same 3-byte hash (what zlib's head[]/prev[] would hold when that position is searched)
 *   L2 qzk_lazy_search_kernel  ONE WAVE PER CHUNK, 64 consecutive positions at a time, every lane walking its own
 *                              position's chain: longest_match() as zlib runs it with a fresh best_len, once for the
 *                              full chain and once for the quarter chain zlib uses when the held match is already
 *                              `good` (a snapshot of the same walk).  64 pointer chases in flight instead of one.
 *   L3 qzk_lazy_parse_kernel   the lazy evaluation itself, serial per chunk but cheap: a table lookup per step.
 *
 * Why the stored answers are exact although zlib starts a search from best_len = prev_length: a search only runs when
 * prev_length < max_lazy <= nice_match, updates happen at strict improvements only, and the walk stops at the first
 * improvement that reaches nice_match - so whenever the search would return more than prev_length it returns the same
 * (length, start) as the search that started from 2, and otherwise its answer is not used.
 * Window slides (chunks above 64 KB) change nothing in the candidate sets (what a slide drops is already beyond
 * MAX_DIST) except for the position that becomes window position 0 == NIL; where the window origin is at a given parse
 * point is a function of the position alone (qzk_lazy_state).
 */
#ifndef QZK_DEFLATE_LAZY_H
#define QZK_DEFLATE_LAZY_H
#include "qzk_deflate_lz77_lane.h"

typedef struct { uint32_t f, q; } qzk_lazyres;     /* len << 16 | dist for the full / quarter chain; 0 = nothing of 3 or more */

/* what a hop of the search needs from a candidate, in ONE 16-byte access: the link to the next candidate (distance to
 * the previous position with the same hash, 0 = none) and the candidate's first 14 bytes - most candidates differ
 * from the string searched for within those, and then no second (scattered) access is needed for the compare */
#define QZK_REC_BYTES 14
typedef struct __attribute__((aligned(16))) { uint32_t w[4]; } qzk_lazyrec;    /* w[0] = pd | b0 << 16 | b1 << 24, then b2.. */
QZ_DEV uint32_t qzk_rec_pd(const qzk_lazyrec &r) { return r.w[0] & 0xffff; }
/* common prefix of the two records' byte strings, 0..14 */
QZ_DEV int qzk_rec_common(const qzk_lazyrec &a, const qzk_lazyrec &b)
{
    uint32_t x = (a.w[0] ^ b.w[0]) >> 16;
    if (x) return qz_ctz32(x) >> 3;
    x = a.w[1] ^ b.w[1]; if (x) return 2 + (qz_ctz32(x) >> 3);
    x = a.w[2] ^ b.w[2]; if (x) return 6 + (qz_ctz32(x) >> 3);
    x = a.w[3] ^ b.w[3]; if (x) return 10 + (qz_ctz32(x) >> 3);
    return QZK_REC_BYTES;
}

/* zlib's window bookkeeping as of a loop top at chunk offset p: origin of the window and how far it is filled */
QZ_DEV void qzk_lazy_state(uint32_t p, uint32_t n, uint32_t *base_out, uint32_t *fill_out)
{
    uint32_t base = 0, fill = n < 65536u ? n : 65536u, avail = n - fill;
    for (;;) {
        if (fill > p && fill - p >= QZK_MINLOOK) break;
        bool did = false;
        if (p - base >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST)) { base += QZK_WSIZE; did = true; }
        if (avail) {
            const uint32_t more = 65536u - (fill - base), rd = avail < more ? avail : more;
            fill += rd; avail -= rd; did = did || rd != 0;
        }
        if (!did) break;
    }
    *base_out = base; *fill_out = fill;
}

/* L1: one wave per chunk, 64 positions a trip.  A position's predecessor is the nearest earlier lane of the trip with
 * its hash, else what the table holds from earlier trips; the last lane of each hash writes the table.  The lanes are
 * grouped hash by hash (one ballot per distinct hash of the trip).
 * head_all: 65536 words per chunk, zeroed by the host, holds position + 1.  pd_all: chunk_sz entries per chunk. */
QZ_KERNEL_MAX(64) qzk_lazy_chain_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                                        const uint32_t *cdesc, uint32_t *head_all, qzk_lazyrec *rec_all)
{
    const uint32_t chunk = blockIdx.x;
    if (chunk >= nchunks) return;
    const int lane = qz_lane();
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    const uint8_t *in = src + coff;
    uint32_t *head = head_all + (uint64_t)chunk * QZK_HSIZE;
    qzk_lazyrec *rec = rec_all + coff;
    for (uint32_t P0 = 0; P0 + 3 <= n; P0 += 64) {
        const uint32_t p = P0 + (uint32_t)lane;
        const bool valid = p + 3 <= n;
        uint32_t h = 0;
        if (valid) h = (((uint32_t)(in[p] & 0xf) << 12) ^ ((uint32_t)in[p + 1] << 6) ^ in[p + 2]) & 0xffff;
        int near = -1; bool later = false;
        uint64_t todo = qz_ballot(valid);
        while (todo) {
            const int j = qz_ctz64(todo);
            const uint32_t hj = qz_readlane(h, j);
            const uint64_t grp = qz_ballot(valid && h == hj);
            if ((grp >> lane) & 1) {
                const uint64_t below = grp & qz_below(lane);
                near = below ? qz_msb64(below) : -1;
                later = (grp >> lane) >> 1 != 0;
            }
            todo &= ~grp;
        }
        qz_wave_sync();                                 /* the previous trip's table stores are visible */
        uint32_t q1 = 0;                                /* predecessor's position + 1, 0 = none */
        if (valid) q1 = near >= 0 ? P0 + (uint32_t)near + 1 : head[h];
        if (valid) {
            uint64_t lo = 0, hi = 0;                    /* bytes p .. p+13 (zero past the end of the chunk: lengths are clamped anyway) */
            if (p + 16 <= n) { lo = ((const qzk_u64u *)(in + p))->v; hi = ((const qzk_u64u *)(in + p + 8))->v; }
            else for (uint32_t k = 0; k < QZK_REC_BYTES && p + k < n; k++) { if (k < 8) lo |= (uint64_t)in[p + k] << (8 * k); else hi |= (uint64_t)in[p + k] << (8 * (k - 8)); }
            qzk_lazyrec r;
            r.w[0] = ((q1 != 0 && p + 1 - q1 <= 32767u) ? p + 1 - q1 : 0) | ((uint32_t)lo << 16);
            r.w[1] = (uint32_t)(lo >> 16); r.w[2] = (uint32_t)(lo >> 48) | ((uint32_t)hi << 16); r.w[3] = (uint32_t)(hi >> 16);
            rec[p] = r;
        }
        qz_wave_sync();                                 /* every lane has read the table before any lane writes it */
        if (valid && !later) head[h] = p + 1;
    }
}

/* common prefix of in[a..] and in[b..] (b < a), at most maxlen; both stay inside the chunk */
QZ_DEV int qzk_lazy_matchlen(const uint8_t *in, uint32_t a, uint32_t b, int maxlen)
{
    int len = 0;
    while (len + 8 <= maxlen) {
        const uint64_t x = ((const qzk_u64u *)(in + a + len))->v ^ ((const qzk_u64u *)(in + b + len))->v;
        if (x) return len + (__builtin_ctzll(x) >> 3);
        len += 8;
    }
    while (len < maxlen && in[a + len] == in[b + len]) len++;
    return len;
}

/* L2: one wave per chunk.  Every lane works on one position's chain, one candidate per trip of the loop; a lane that
 * is done with its position takes the next unsearched one, so the wave's time is the chunk's total number of hops
 * over 64, not the sum of every 64 positions' longest chain.  The link to the next candidate is loaded before the
 * compare that decides whether it is needed: one memory round trip per hop instead of two. */
QZ_KERNEL_MAX(64) qzk_lazy_search_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                                         const uint32_t *cdesc, const qzk_lazyrec *rec_all, qzk_lazyres *res_all, qzk_lvlcfg cfg)
{
    const uint32_t chunk = blockIdx.x;
    if (chunk >= nchunks) return;
    const int lane = qz_lane();
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    const uint8_t *in = src + coff;
    const qzk_lazyrec *rec = rec_all + coff;
    qzk_lazyres *res = res_all + coff;
    const int quarter = cfg.chain >> 2;
    uint32_t next = 0;                                  /* first position nobody has taken yet (wave-uniform) */
    bool active = false;
    uint32_t p = 0, cur = 0, lim = 0, bd = 0, rq = 0;
    qzk_lazyrec own; own.w[0] = own.w[1] = own.w[2] = own.w[3] = 0;
    int maxlen = 0, nice = 0, best = 2, hops = 0;
    bool snapped = false;
    for (;;) {
        const uint64_t idle = qz_ballot(!active);
        if (idle && next < n) {
            const uint32_t mine = next + (uint32_t)qz_popc64(idle & qz_below(lane));
            if (!active && mine < n) {
                p = mine;
                uint32_t base, fill;
                qzk_lazy_state(p, n, &base, &fill);
                const uint32_t la = fill - p;
                if (la >= 3) own = rec[p];
                const uint32_t d0 = la >= 3 ? qzk_rec_pd(own) : 0;
                if (d0 != 0 && d0 <= (uint32_t)QZK_MAXDIST && p - d0 > base) {       /* hash_head != NIL, within MAX_DIST */
                    maxlen = la < 258 ? (int)la : 258; nice = la < (uint32_t)cfg.nice ? (int)la : cfg.nice;
                    lim = p > (uint32_t)QZK_MAXDIST + base ? p - QZK_MAXDIST : base;  /* chained candidates: strictly above */
                    cur = p - d0; bd = 0; best = 2; hops = 0; snapped = false; rq = 0;
                    active = true;
                } else {
                    qzk_lazyres z; z.f = 0; z.q = 0;
                    res[p] = z;
                }
            }
            next += (uint32_t)qz_popc64(idle);
        }
        if (qz_ballot(active) == 0) { if (next >= n) break; continue; }
        if (active) {
            const qzk_lazyrec c = rec[cur];            /* link and first bytes of the candidate in one access */
            const uint32_t d = qzk_rec_pd(c);
            int len = qzk_rec_common(own, c);
            if (len == QZK_REC_BYTES && maxlen > QZK_REC_BYTES)
                len += qzk_lazy_matchlen(in, p + QZK_REC_BYTES, cur + QZK_REC_BYTES, maxlen - QZK_REC_BYTES);
            if (len > maxlen) len = maxlen;
            hops++;
            bool stop = false;
            if (len > best) { best = len; bd = p - cur; stop = len >= nice; }
            if (!stop) {
                stop = d == 0 || cur < d || cur - d <= lim || hops >= cfg.chain;
                cur -= d;
            }
            if (!snapped && (hops == quarter || stop)) { snapped = true; rq = best >= 3 ? ((uint32_t)best << 16) | bd : 0; }
            if (stop) {
                qzk_lazyres r; r.f = best >= 3 ? ((uint32_t)best << 16) | bd : 0; r.q = rq;
                res[p] = r;
                active = false;
            }
        }
    }
}

/* L3: one wave per chunk, the parse itself wave-uniform (scalar registers).  zlib's deflate_slow loop with the searches
 * replaced by lookups; the window bookkeeping at the loop top is zlib's own (fill_window), as in K1b.  The stored
 * answers and the input bytes of 64 positions sit in the lanes (one coalesced load per 64 positions, a cross-lane read
 * per step), and symbols leave 64 at a time. */
QZ_KERNEL_MAX(64) qzk_lazy_parse_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                                        const uint32_t *cdesc, const qzk_lazyres *res_all, uint8_t *sym_lc, uint16_t *sym_dist,
                                        qzk_lzmeta *meta, qzk_lvlcfg cfg)
{
    const uint32_t chunk = blockIdx.x;
    if (chunk >= nchunks) return;
    const int lane = qz_lane();
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    const uint8_t *in = src + coff;
    const qzk_lazyres *res = res_all + coff;
    uint8_t *olc = sym_lc + coff;
    uint16_t *odist = sym_dist + coff;
    qzk_lzmeta *mt = meta + chunk;

    uint32_t base = 0, fill = n < 65536u ? n : 65536u, avail_in = n - fill, pos = 0;
    uint32_t nsym = 0, nfull = 0, cur_bstart = 0, can_store = 0, inblock = 0;
    uint32_t match_len = 2, match_at = 0;
    bool held = false;
    mt->bstart[0] = 0;                                  /* wave-uniform values: every lane stores the same word */
    /* lanes: answers for positions W0 + lane, input byte of position W0 - 1 + lane (the byte a literal step emits) */
    uint32_t W0 = 0, wf = 0, wq = 0, wb = 0;
    bool have = false;
    /* symbols waiting to be stored: lane k holds the k-th of them */
    uint32_t sb_lc = 0, sb_d = 0, sbn = 0, stored = 0;
#define QZK_LAZY_FLUSH() do { if ((uint32_t)lane < sbn) { olc[stored + (uint32_t)lane] = (uint8_t)sb_lc; odist[stored + (uint32_t)lane] = (uint16_t)sb_d; } \
        stored += sbn; sbn = 0; } while (0)
#define QZK_LAZY_TALLY(lc, d, full) do { const uint32_t lc_ = (lc), d_ = (d); if ((uint32_t)lane == sbn) { sb_lc = lc_; sb_d = d_; } \
        sbn++; nsym++; if (sbn == 64) QZK_LAZY_FLUSH(); (full) = ++inblock == QZK_LITBUF; } while (0)
#define QZK_LAZY_CLOSE(nb) do { if (cur_bstart >= base) can_store |= 1u << nfull; \
        nfull++; inblock = 0; cur_bstart = (nb); if (nfull < QZK_MAXBLK) mt->bstart[nfull] = (nb); } while (0)
    for (;;) {
        uint32_t look = fill - pos;
        if (look < QZK_MINLOOK) {                       /* zlib fill_window() */
            if (pos - base >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST)) base += QZK_WSIZE;
            if (avail_in) {
                uint32_t more = 65536u - (fill - base), rd = avail_in < more ? avail_in : more;
                fill += rd; avail_in -= rd;
            }
            look = fill - pos;
            if (look == 0) break;
        }
        if (!have || pos >= W0 + 64) {                  /* next 64 positions into the lanes */
            W0 = pos; have = true;
            const uint32_t q = W0 + (uint32_t)lane;
            qzk_lazyres r; r.f = 0; r.q = 0;
            if (q < n) r = res[q];
            wf = r.f; wq = r.q;
            wb = (q >= 1 && q - 1 < n) ? in[q - 1] : 0;
        }
        const int wi = (int)(pos - W0);
        const uint32_t prev_len = match_len, prev_at = match_at;
        uint32_t mlen = 2;
        if (look >= 3 && prev_len < (uint32_t)cfg.lazy) {
            const uint32_t v = prev_len >= (uint32_t)cfg.good ? qz_readlane(wq, wi) : qz_readlane(wf, wi);
            if ((v >> 16) > prev_len) {
                mlen = v >> 16; match_at = pos - (v & 0xffff);
                if (mlen > look) mlen = look;
                if (mlen == 3 && pos - match_at > 4096u) mlen = 2;          /* TOO_FAR */
            }
        }
        bool full;
        if (prev_len >= 3 && mlen <= prev_len) {        /* the held match stands */
            QZK_LAZY_TALLY(prev_len - 3, pos - 1 - prev_at, full);
            pos += prev_len - 1;
            held = false; match_len = 2;
            if (full) QZK_LAZY_CLOSE(pos);
        } else {
            if (held) {                                 /* byte pos-1 goes out as a literal */
                QZK_LAZY_TALLY(qz_readlane(wb, wi), 0, full);
                if (full) QZK_LAZY_CLOSE(pos);
            }
            held = true; match_len = mlen;
            pos++;
        }
    }
    if (held) { bool full; const uint32_t lastb = in[pos - 1]; QZK_LAZY_TALLY(lastb, 0, full); (void)full; }
    QZK_LAZY_FLUSH();
    if (cur_bstart >= base) can_store |= 1u << nfull;
    mt->nsym = nsym; mt->nfull = nfull; mt->can_store = can_store; mt->n = n;
#undef QZK_LAZY_TALLY
#undef QZK_LAZY_CLOSE
#undef QZK_LAZY_FLUSH
}

#endif
