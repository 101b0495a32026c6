/*
 * qzk_deflate_wide.h — K1w: zlib-exact greedy LZ77 parse ("deflate_fast", level 1) of one chunk of at most 64 KB per
 * WORKGROUP, 1024 positions at a time, gfx950.  EXPERIMENTAL (QATZIP_AMD_K1=wide): the round-3 direction of DESIGN.md
 * section 7, built to be exact first; the product path is qzk_deflate_lz77.h.
 *
 * Why: K1 hides the serial parse behind 64-position windows and pays with a 1 MiB candidate table per chunk in flight -
 * 4096 of them, HBM-resident, one random 64-byte fetch per position (98 % of the kernel's traffic).  With ONE chunk per
 * CU the table is zlib's own and fits on chip: head[65536] as 16-bit positions in LDS (128 KiB), prev[] 128 KiB per
 * workgroup in global memory that never leaves the L2.  That only works if the parse inside a chunk is found in parallel
 * over wide windows, and tools/k1_fixpoint_model.py says it can be: assume every position of the window gets inserted,
 * match all 1024 positions at once, parse, derive the inserted set the parse implies, repeat - zlib's own parse is the
 * fixpoint (by induction over the positions), reached after 2-4 rounds, and under 1.1 % of the positions ever change.
 *
 * A window, all 1024 threads (thread t = position ws + t):
 *   1. sort (hash << 10 | t) in LDS: a position's in-window candidates are its predecessors in its hash group;
 *   2. the (exact) candidates from before the window: head[h] and up to three prev[] links;
 *   3. rounds: every thread walks its candidates in zlib's order - in-window predecessors that are inserted under the
 *      current assumption, newest first, then the table's - with zlib's rules (chain of 4, nice 8, NIL = 0, MAX_DIST);
 *      the parse points are found by pointer jumping over next[p] = p + max(1, len); the inserted set of that parse
 *      replaces the assumption; until nothing changes;
 *   4. symbols (wave ballots + a scan of the wave totals), the 32767-symbol block marks, and the chains: an inserted
 *      position links to the previous inserted one of its hash group (or the old head), the group's last becomes head.
 * Same symbol / meta contract as K1 (qzk_lzmeta), so K2 and everything after it are shared.
 */
#ifndef QZK_DEFLATE_WIDE_H
#define QZK_DEFLATE_WIDE_H
#include "qzk_deflate_lz77.h"

#define QZW_W 1024
#define QZW_LIM (QZW_W - 4)        /* parse points per window: the interiors of a match of at most 4 stay inside it */
#define QZW_NOHASH 0x10000u        /* sort key of a position with fewer than three bytes ahead: behind every hash */

typedef struct {
    uint16_t head[65536];          /* most recent inserted position per hash, 0 = NIL (position 0 is zlib's NIL) */
    uint32_t keys[QZW_W];          /* (hash << 10 | t), sorted */
    uint16_t rank[QZW_W];          /* t -> its place in keys[] */
    uint16_t jump[2][QZW_W];       /* pointer jumping, ping-pong: window index after 2^s hops (QZW_W = outside) */
    uint16_t mlen[QZW_W], mdist[QZW_W];
    uint8_t mark[2][QZW_W];        /* reachable from the window start within 2^s hops */
    uint8_t ins[2][QZW_W];         /* inserted: the assumption of a round / what its parse implies */
    uint32_t wtot[16];             /* parse points per wave */
    uint32_t u[8];                 /* workgroup-uniform mailboxes */
} qzw_lds;

QZ_DEV uint32_t qzw_matchlen(const uint8_t *in, uint32_t p, uint32_t q, uint32_t maxlen)
{
    uint32_t len = 0;
    while (len + 4 <= maxlen) {
        const uint32_t x = qz_ld32(in + p + len) ^ qz_ld32(in + q + len);
        if (x) return len + ((uint32_t)qz_ctz32(x) >> 3);
        len += 4;
    }
    while (len < maxlen && in[p + len] == in[q + len]) len++;
    return len;
}

QZ_DEV void qzw_chunk(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t chunk, uint8_t *olc, uint16_t *odist,
                      qzk_lzmeta *meta, uint16_t *prev, const uint32_t *cdesc, qzw_lds *L)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    const uint8_t *in = src + coff;
    qzk_lzmeta *mt = meta + chunk;

    for (uint32_t i = tid; i < 32768; i += QZW_W) ((uint32_t *)L->head)[i] = 0;
    uint32_t ws = 0, nsym = 0, nfull = 0, cur_bstart = 0, can_store = 0, base = 0;      /* workgroup-uniform */
    mt->bstart[0] = 0;
    qz_block_sync();

    while (ws < n) {
        const uint32_t p = ws + tid;
        const uint32_t avail = p < n ? n - p : 0;
        const bool canh = avail >= 3;
        const uint32_t nvalid = n - ws < QZW_W ? n - ws : QZW_W;
        const uint32_t lim = nvalid < QZW_LIM ? nvalid : QZW_LIM;
        const uint32_t maxlen = avail < 258 ? avail : 258, nice = avail < QZK_NICE ? avail : QZK_NICE;
        uint32_t h = 0;
        if (canh) { const uint32_t w = qzk_ld32g(src, coff + p, src_len); h = (((w & 0xf) << 12) ^ (((w >> 8) & 0xff) << 6) ^ ((w >> 16) & 0xff)) & 0xffff; }

        /* ---- 1. the window's hash groups ---- */
        L->keys[tid] = ((canh ? h : QZW_NOHASH) << 10) | tid;
        qz_block_sync();
        for (uint32_t k = 2; k <= QZW_W; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const uint32_t o = tid ^ j;
                if (o > tid) {
                    const uint32_t a = L->keys[tid], b = L->keys[o];
                    if ((a > b) == ((tid & k) == 0)) { L->keys[tid] = b; L->keys[o] = a; }
                }
                qz_block_sync();
            }
        L->rank[L->keys[tid] & 1023] = (uint16_t)tid;
        qz_block_sync();
        const uint32_t r = L->rank[tid];

        /* ---- 2. the candidates from before the window (exact: earlier windows are final) ---- */
        uint32_t qc0 = 0, qc1 = 0, qc2 = 0, qc3 = 0;
        if (canh) {
            qc0 = L->head[h];
            if (qc0) qc1 = prev[qc0];
            if (qc1) qc2 = prev[qc1];
            if (qc2) qc3 = prev[qc2];
        }
        const uint32_t lo = p > QZK_MAXDIST ? p - QZK_MAXDIST : 0;        /* chained candidates must lie above zlib's limit */

        /* ---- 3. rounds ---- */
        int cur = 0, rounds = 0;
        L->ins[0][tid] = canh ? 1 : 0;
        qz_block_sync();
        for (;;) {
            uint32_t best = 2, bq = 0;
            if (canh) {
                uint32_t cnt = 0; bool done = false;
                for (uint32_t i = 1; i <= r && cnt < 4 && !done; i++) {             /* in-window, newest first */
                    const uint32_t k = L->keys[r - i];
                    if ((k >> 10) != h) break;
                    const uint32_t t = k & 1023;
                    if (!L->ins[cur][t]) continue;
                    const uint32_t q = ws + t;
                    if (q == 0) { done = true; break; }                             /* NIL ends the chain */
                    cnt++;
                    const uint32_t len = qzw_matchlen(in, p, q, maxlen);
                    if (len > best) { best = len; bq = q; }
                    if (len >= nice) done = true;
                }
                for (uint32_t i = 0; i < 4 && cnt < 4 && !done; i++) {              /* then the table's */
                    const uint32_t q = i == 0 ? qc0 : i == 1 ? qc1 : i == 2 ? qc2 : qc3;
                    if (q == 0) break;
                    if (cnt == 0 ? p - q > QZK_MAXDIST : q <= lo) break;            /* head: dist <= MAX_DIST; chained: > limit */
                    cnt++;
                    const uint32_t len = qzw_matchlen(in, p, q, maxlen);
                    if (len > best) { best = len; bq = q; }
                    if (len >= nice) done = true;
                }
            }
            const uint32_t ml = best >= 3 ? best : 0;
            L->mlen[tid] = (uint16_t)ml; L->mdist[tid] = (uint16_t)(ml ? p - bq : 0);
            {
                const uint32_t nx = tid + (ml ? ml : 1);
                L->jump[0][tid] = (uint16_t)(nx < QZW_W ? nx : QZW_W);
                L->mark[0][tid] = tid == 0;
            }
            qz_block_sync();
            int c = 0;
            for (int s = 0; s < 10; s++) {                                          /* reachable within 2^(s+1) hops */
                const uint32_t j = L->jump[c][tid];
                const uint8_t m = L->mark[c][tid];
                L->mark[c ^ 1][tid] = m;
                L->jump[c ^ 1][tid] = (uint16_t)(j < QZW_W ? L->jump[c][j] : QZW_W);
                qz_block_sync();
                if (m && j < QZW_W) L->mark[c ^ 1][j] = 1;
                qz_block_sync();
                c ^= 1;
            }
            /* parse points of this round: reachable and below the window's limit (c == 0 again after ten steps) */
            const bool pp = L->mark[c][tid] && tid < lim;
            L->mark[1][tid] = pp;                                                   /* kept for the neighbours and for step 4 */
            if (tid == 0) L->u[0] = 0;
            qz_block_sync();
            bool insd = pp && canh;
            for (uint32_t b = 1; b <= 3 && !insd; b++) {                            /* interior of a short match b positions back */
                if (tid < b) break;
                const uint32_t t = tid - b, m = L->mlen[t];
                if (L->mark[1][t] && m > b && m >= 3 && m <= QZK_MAXINS && (n - (ws + t)) - m >= 3) insd = true;
            }
            L->ins[cur ^ 1][tid] = insd ? 1 : 0;
            if ((insd ? 1 : 0) != L->ins[cur][tid]) L->u[0] = 1;
            qz_block_sync();
            const uint32_t changed = L->u[0];
            qz_block_sync();
            cur ^= 1;
            if (!changed || ++rounds > 2 * QZW_W) break;           /* (every round finalises at least one more parse point) */
        }

        /* ---- 4. the window is final: symbols, block marks, chains ---- */
        const bool pp = L->mark[1][tid] != 0;
        const uint32_t ml = L->mlen[tid], md = L->mdist[tid];
        const uint32_t step = ml ? ml : 1;
        const uint64_t ppm = qz_ballot(pp);
        if (lane == 0) L->wtot[wv] = (uint32_t)qz_popc64(ppm);
        if (tid == 0) { L->u[1] = 0; L->u[2] = 0xffffffffu; L->u[3] = 0xffffffffu; }
        qz_block_sync();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < QZW_W / 64; w++) { const uint32_t t = L->wtot[w]; if (w < wv) before += t; total += t; }
        const uint32_t idx = nsym + before + (uint32_t)qz_popc64(ppm & qz_below((int)lane));
        if (pp) {
            olc[idx] = (uint8_t)(ml ? ml - 3 : in[p]);
            odist[idx] = (uint16_t)md;
            if (tid + step >= lim) L->u[1] = tid + step;                            /* the last parse point: where the next window starts */
            if ((idx + 1) % QZK_LITBUF == 0) { L->u[2] = p + step; L->u[3] = p; }   /* completes a block (at most one per window) */
            /* zlib slides its window at the first loop top with less than MIN_LOOKAHEAD ahead and strstart >= 65274 */
            if (p >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST) && n - p < QZK_MINLOOK) atomicMin(&L->u[4], p);
        }
        /* chains: previous inserted position of my hash group, and whether I am its last */
        const bool mine = L->ins[cur][tid] != 0;
        uint32_t link = 0; bool lastg = false;
        if (mine) {
            link = L->head[h];
            for (uint32_t i = 1; i <= r; i++) {
                const uint32_t k = L->keys[r - i];
                if ((k >> 10) != h) break;
                if (L->ins[cur][k & 1023]) { link = ws + (k & 1023); break; }
            }
            lastg = true;
            for (uint32_t i = r + 1; i < QZW_W; i++) {
                const uint32_t k = L->keys[i];
                if ((k >> 10) != h) break;
                if (L->ins[cur][k & 1023]) { lastg = false; break; }
            }
        }
        qz_block_sync();
        if (mine) { prev[p] = (uint16_t)link; if (lastg) L->head[h] = (uint16_t)p; }
        const uint32_t adv = L->u[1], nb = L->u[2], closer = L->u[3];
        const uint32_t slide_at = L->u[4];
        if (nb != 0xffffffffu) {
            if (closer >= slide_at) base = QZK_WSIZE;
            if (cur_bstart >= base) can_store |= 1u << nfull;
            nfull++;
            cur_bstart = nb;
            if (nfull < QZK_MAXBLK) mt->bstart[nfull] = nb;
        }
        if (slide_at != 0xffffffffu) base = QZK_WSIZE;
        nsym += total;
        ws += adv;
        qz_block_sync();
    }
    /* zlib's final loop top (lookahead == 0) may still slide before the last flush */
    if (n >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST)) base = QZK_WSIZE;
    if (cur_bstart >= base) can_store |= 1u << nfull;
    mt->nsym = nsym; mt->nfull = nfull; mt->can_store = can_store; mt->n = n;         /* uniform, every thread stores the same */
}

/* persistent workgroups, one per CU: each pulls chunk numbers; prev[] of workgroup g at prevtab + g * 65536 */
QZ_KERNEL_MAX(QZW_W) qzk_lz77_wide_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                                          uint8_t *sym_lc, uint16_t *sym_dist, qzk_lzmeta *meta, uint16_t *prevtab,
                                          uint32_t *counter, const uint32_t *cdesc)
{
    QZ_LDS qzw_lds L;
    uint16_t *prev = prevtab + (size_t)blockIdx.x * 65536;
    for (;;) {
        if (threadIdx.x == 0) { L.u[5] = atomicAdd(counter, 1u); L.u[4] = 0xffffffffu; }
        qz_block_sync();
        const uint32_t chunk = L.u[5];
        qz_block_sync();
        if (chunk >= nchunks) break;
        const uint64_t coff = (uint64_t)chunk * chunk_sz;
        qzw_chunk(src, src_len, chunk_sz, chunk, sym_lc + coff, sym_dist + coff, meta, prev, cdesc, &L);
    }
}

#endif
