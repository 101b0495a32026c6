/*
 * qzk_deflate_wide.h — K1w: zlib-exact greedy LZ77 parse ("deflate_fast", level 1) of one chunk of at most 64 KB per
 * WORKGROUP (1024 threads = the sixteen waves of a CU), 1024 positions at a time, everything it looks up ON CHIP, gfx950.
 * The latency path: a launch of few chunks (a lone qzCompress of one block, the requests a few threads have in flight)
 * gives every chunk a whole CU instead of one wave (qzk_deflate_lz77.h: ~3.5 ms a chunk) - and the form north_star
 * describes: one chunk per workgroup, history window and tables in LDS.
 *
 * What it replaces: zlib's deflate_fast()/longest_match() (src/qatzip_sw.c:197; CPU restatement oracle/qzo_deflate.c).
 * Same symbol / meta contract as K1 (qzk_lzmeta), so K2 and everything after it are shared.
 *
 * On chip: the chunk itself (64 KiB of LDS: every compare reads it), zlib's prev[] (32768 x u16, LDS), head[] as 65536 x
 * u16 per workgroup in global memory that lives in the L2 (one 16-bit gather per hash GROUP of a window, issued before
 * the window's sort and back long before it is needed).  HBM sees the input once and the symbols once.
 *
 * The parse is serial by definition - which positions enter the hash chains depends on the matches taken - so a window
 * is solved as a FIXPOINT (tools/k1_fixpoint_model.py; tools/k1_prefix_model.c): assume an inserted set, give every
 * position the candidates that set implies, match, parse, derive the inserted set that parse implies, repeat.  Each
 * round ends with a check that needs no compare: the candidate lists under the derived set against the lists the round
 * used.  The parse points before the first one whose list changed are FINAL (induction over the positions: the first
 * parse point never has an in-window candidate), so a window that has not converged after QZX_RMAX rounds is simply cut
 * there and the next window starts at that parse point - no bound on rounds is needed for exactness, and every window
 * finalises at least its first parse point.
 *
 * One window, thread = position (ws + tid) in some phases and = sorted element in others:
 *   1. hash, the 16-bit gather of head[h] (L2), a stable two-pass radix sort of (hash << 10 | t) in LDS: a position's
 *      in-window candidates are the INSERTED ones among its predecessors in its hash group;
 *   2. in sorted order: the group structure as ballots (a group's members sit in neighbouring lanes), the chain
 *      head[h] -> prev[] -> prev[] -> prev[] from LDS, 16 speculative bytes of every candidate compared out of LDS;
 *   3. rounds: the four newest inserted members of the group before me = four bit scans of a ballot (+ a carry list for a
 *      group that began in an earlier wave) -> zlib's rules (chain of 4, nice 8, NIL, MAX_DIST) -> matches that outgrow the
 *      16 bytes are extended by whole waves, one (position, distance) run at a time -> pointer doubling inside every wave
 *      (where does the parse leave the wave from each of its 64 positions, which positions does it touch on the way) ->
 *      the sixteen exit tables chained -> parse points, inserted set -> the check;
 *   4. final prefix: symbols (ballot ranks + the waves' totals), block marks, prev[] links, the groups' new heads.
 */
#ifndef QZK_DEFLATE_WIDE_H
#define QZK_DEFLATE_WIDE_H
#include "qzk_deflate_lz77.h"

#define QZX_W 1024
#define QZX_NW (QZX_W / 64)
#define QZX_CAP 16                 /* speculative compare depth (>= nice_match) */
#ifndef QZX_RMAX
#define QZX_RMAX 3                 /* rounds before a window is cut at its final prefix */
#endif
#define QZX_PAD 352                /* zero bytes behind the chunk: compares and extensions read past its end */
#define QZX_NOPOS 0xffffu
/* Phase clocks, a run-time option (xprof != NULL, QATZIP_AMD_WIDE_PROF=1): WAVE 0 keeps them in LDS - the whole wave takes
 * the (scalar) branch and stores the same words; a lane-0-only block between cross-lane operations invites the compiler
 * to thread lane 0 past them (DESIGN.md K1 pitfall 1) */
#ifndef QZ_SIM
#define QZX_T(k) do { if (xprof && (threadIdx.x >> 6) == 0) { const uint64_t t_ = __builtin_readcyclecounter(), p_ = L->xp[15]; const uint64_t a_ = L->xp[k]; L->xp[k] = a_ + (t_ - p_); L->xp[15] = t_; } } while (0)
#define QZX_C(k, v) do { if (xprof && (threadIdx.x >> 6) == 0) { const uint64_t a_ = L->xp[k]; L->xp[k] = a_ + (uint64_t)(v); } } while (0)
#else
#define QZX_T(k) do { } while (0)
#define QZX_C(k, v) do { } while (0)
#endif
#if defined(QZ_SIM) && defined(QZX_DEBUG)
#define QZX_DBG(...) do { if (threadIdx.x == 0) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)
#else
#define QZX_DBG(...) do { } while (0)
#endif

typedef struct { uint32_t cnt, cont, hlast, e[4]; uint32_t pad; } qzx_pub;      /* 32 bytes: a wave's last hash group */

typedef struct {
    uint32_t in32[(65536 + QZX_PAD) / 4];      /* the chunk */
    uint16_t prev[32768];                      /* zlib's prev[]: previous inserted position with the same hash, by position & 0x7fff */
    uint32_t keyA[QZX_W];                      /* sort ping; afterwards the sorted keys (hash << 10 | t) */
    union {
        uint32_t keyB[QZX_W];                  /* sort pong */
        struct { uint16_t mlen[QZX_W], mdist[QZX_W]; } m;      /* by position: match length (bit 15: capped at 16) / distance */
    };
    union {
        struct { uint16_t cnt[QZX_NW][256]; uint16_t base[256]; uint16_t wsum[8]; } s;     /* radix pass */
        struct { uint16_t pre0[QZX_W]; uint16_t exitp[QZX_W]; } w;   /* after the sort */
    };
    qzx_pub pub[QZX_NW];
    uint64_t insw[QZX_NW], ppw[QZX_NW];        /* by position: inserted set / parse points of the round, a word per wave */
    uint32_t spill[QZX_NW + 1];                /* interiors of a short match that fall into the next wave (3 bits) */
    uint32_t wtot[QZX_NW];
    uint32_t u[16];                            /* workgroup-uniform mailboxes */
    uint64_t xp[16];                           /* cycles per phase when asked for ([15]: the running clock) */
} qzx_lds;

/* 16 bytes at byte offset a of the chunk (LDS), little endian, any alignment */
QZ_DEV void qzx_ld16(const uint32_t *in32, uint32_t a, uint32_t *o)
{
    const uint32_t i = a >> 2, s = a & 3;
    const uint32_t d0 = in32[i], d1 = in32[i + 1], d2 = in32[i + 2], d3 = in32[i + 3], d4 = in32[i + 4];
    o[0] = qzk_alignbyte(d1, d0, s); o[1] = qzk_alignbyte(d2, d1, s); o[2] = qzk_alignbyte(d3, d2, s); o[3] = qzk_alignbyte(d4, d3, s);
}
QZ_DEV uint32_t qzx_ld4(const uint32_t *in32, uint32_t a)
{
    const uint32_t i = a >> 2;
    return qzk_alignbyte(in32[i + 1], in32[i], a & 3);
}
/* common prefix of two 16-byte strings, 0..16 */
QZ_DEV uint32_t qzx_len16(const uint32_t *a, const uint32_t *b)
{
    uint32_t len = QZX_CAP, d;
    d = a[3] ^ b[3]; if (d) len = 12 + ((uint32_t)qz_ctz32(d) >> 3);
    d = a[2] ^ b[2]; if (d) len = 8 + ((uint32_t)qz_ctz32(d) >> 3);
    d = a[1] ^ b[1]; if (d) len = 4 + ((uint32_t)qz_ctz32(d) >> 3);
    d = a[0] ^ b[0]; if (d) len = ((uint32_t)qz_ctz32(d) >> 3);
    return len;
}
QZ_DEV uint32_t qzx_rank(uint64_t m, int lane) { return (uint32_t)qz_popc64(m & qz_below(lane)); }
QZ_DEV uint64_t qzx_shfl64(uint64_t v, int src)
{
    return (uint64_t)qz_shfl((uint32_t)v, src) | ((uint64_t)qz_shfl((uint32_t)(v >> 32), src) << 32);
}
QZ_DEV uint64_t qzx_readlane64(uint64_t v, int src)
{
    return (uint64_t)qz_readlane((uint32_t)v, src) | ((uint64_t)qz_readlane((uint32_t)(v >> 32), src) << 32);
}
QZ_DEV uint16_t qzx_ld_head(const uint16_t *p)
{
#ifdef QZ_SIM
    return *p;
#else
    /* served by the L2: the head[] entries are rewritten by other waves of this workgroup window after window, and this
     * CU's vector L1 is not kept coherent with their stores */
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
/* all of this wave's global stores have reached the L2 */
QZ_DEV void qzx_drain_stores()
{
#ifndef QZ_SIM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

/* stable counting pass of the radix sort: src[tid] -> dst[position by (digit, original order)] */
QZ_DEV void qzx_radix_pass(qzx_lds *L, const uint32_t key, const uint32_t shift, uint32_t *dst, const uint32_t tid, const int lane, const uint32_t wv)
{
    const uint32_t d = (key >> shift) & 255u;
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint32_t bit = (d >> b) & 1u;
        const uint64_t bal = qz_ballot(bit != 0);
        m &= bal ^ ((uint64_t)bit - 1ull);
    }
    const uint32_t rank = qzx_rank(m, lane), cnt = (uint32_t)qz_popc64(m);
    {   /* this wave's row of the histogram: zero, then one writer per digit present */
        uint32_t *row = (uint32_t *)L->s.cnt[wv];
        row[2 * lane] = 0; row[2 * lane + 1] = 0;
    }
    qz_lds_sync();
    if (rank == 0) L->s.cnt[wv][d] = (uint16_t)cnt;
    qz_block_sync();
    /* digit totals -> exclusive scan over the digits; per digit the exclusive prefix over the waves */
    uint32_t tot = 0;
    if (tid < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < QZX_NW; w++) { const uint32_t c = L->s.cnt[w][tid]; L->s.cnt[w][tid] = (uint16_t)run; run += c; }
        tot = run;
    }
    uint32_t inc = tot;
    if (tid < 256) {                               /* waves 0-3, wave-uniform */
        for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = qz_shfl(inc, lane - dd); if (lane >= dd) inc += o; }
        if (lane == 63) L->s.wsum[wv] = (uint16_t)inc;
    }
    qz_block_sync();
    if (tid < 256) {
        uint32_t add = 0;
        for (uint32_t w = 0; w < wv; w++) add += L->s.wsum[w];
        L->s.base[tid] = (uint16_t)(add + inc - tot);
    }
    qz_block_sync();
    dst[(uint32_t)L->s.base[d] + (uint32_t)L->s.cnt[wv][d] + rank] = key;
    qz_block_sync();
}

QZ_DEV void qzx_chunk(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t chunk, uint8_t *olc, uint16_t *odist,
                      qzk_lzmeta *meta, uint16_t *head, const uint32_t *cdesc, qzx_lds *L, uint64_t *xprof)
{
    const uint32_t tid = threadIdx.x, wv = tid >> 6;
    const int lane = (int)(tid & 63);
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    qzk_lzmeta *mt = meta + chunk;
    const uint32_t *in32 = L->in32;

#ifndef QZ_SIM
    if (xprof && wv == 0) { if (lane < 15) L->xp[lane] = 0; const uint64_t t_ = __builtin_readcyclecounter(); L->xp[15] = t_; L->xp[14] = t_; }
#endif
    /* ---- the chunk into LDS (zero behind it), head[] cleared ---- */
    for (uint32_t i = tid; i < (65536 + QZX_PAD) / 4; i += QZX_W) {
        uint32_t v = 0;
        if (4 * i < n) { v = qzk_ld32g_fast(src, coff + 4 * i, src_len); if (4 * i + 4 > n) v &= (1u << (8 * (n - 4 * i))) - 1u; }
        L->in32[i] = v;
    }
    {   /* 16 bytes a store, one running pointer (the unrolled form kept 32 addresses alive: 64 registers, spilled - the
         * kernel's 284 bytes of scratch in round 3) */
        typedef uint32_t qzx_u32x4 __attribute__((vector_size(16)));
        const qzx_u32x4 z = {0, 0, 0, 0};
        qzx_u32x4 *hp = (qzx_u32x4 *)head + tid;
#pragma unroll 1
        for (uint32_t i = tid; i < 65536 * 2 / 16; i += QZX_W, hp += QZX_W) *hp = z;
    }
    qzx_drain_stores();
    uint32_t ws = 0, nsym = 0, nfull = 0, cur_bstart = 0, can_store = 0;      /* workgroup-uniform */
    mt->bstart[0] = 0;
    /* zlib slides its window at the first loop top (= parse point) with strstart >= 65274 and less than MIN_LOOKAHEAD
     * ahead: from there on positions at or below 32768 are NIL.  Chunks of at most 64 KB slide at most once. */
    const uint32_t slide_thr = n > (uint32_t)(QZK_WSIZE + QZK_MAXDIST) ?
        ((uint32_t)(QZK_WSIZE + QZK_MAXDIST) > n - (QZK_MINLOOK - 1) ? (uint32_t)(QZK_WSIZE + QZK_MAXDIST) : n - (QZK_MINLOOK - 1)) : 0xffffffffu;
    qz_block_sync();

    QZX_T(12);
    while (ws < n) {
        QZX_C(10, 1);
        const uint32_t base = ws >= slide_thr ? (uint32_t)QZK_WSIZE : 0u;
        uint32_t lim = n - ws < QZX_W ? n - ws : QZX_W;
        if (lim > QZX_W - 3) lim = QZX_W - 3;                  /* the interiors of a short match stay inside the window */
        if (ws < slide_thr && slide_thr - ws < lim) lim = slide_thr - ws;       /* the slide point starts a window of its own */

        QZX_DBG("window ws=%u lim=%u n=%u\n", ws, lim, n);
        /* ================= 1. hash, head gather, sort ================= */
        uint32_t c0raw = 0;
        {
            const uint32_t p = ws + tid;
            const bool canh = p + 3 <= n;
            const uint32_t w0 = qzx_ld4(in32, p < 65536u ? p : 65536u);
            const uint32_t h = canh ? ((((w0 & 0xf) << 12) ^ (((w0 >> 8) & 0xff) << 6) ^ ((w0 >> 16) & 0xff)) & 0xffff) : 0xffffu;
            if (canh) c0raw = qzx_ld_head(head + h);
            const uint32_t key = (h << 10) | tid;
            qzx_radix_pass(L, key, 10, L->keyB, tid, lane, wv);
            QZX_T(0);
            qzx_radix_pass(L, L->keyB[tid], 18, L->keyA, tid, lane, wv);
            QZX_T(1);
        }
        L->w.pre0[tid] = (uint16_t)c0raw;                       /* (the radix scratch is free: the last pass ended with a barrier) */
        if (tid == 0) { L->u[4] = QZX_NOPOS; L->u[5] = QZX_NOPOS; }
        qz_block_sync();

        QZX_DBG(" sorted\n");
        /* ================= 2. sorted order: my element, its group, the chain from before the window ================= */
        const uint32_t key = L->keyA[tid];
        const uint32_t h = key >> 10, t = key & 1023u, p = ws + t;
        const uint32_t avail = p < n ? n - p : 0;
        const bool canh = avail >= 3;
        const uint32_t hshf = qz_shfl(h, lane - 1);
        const uint32_t hprev = lane ? hshf : (tid ? L->keyA[tid - 1] >> 10 : 0xffffffffu);
        const bool segstart = tid == 0 || hprev != h;
        const uint64_t S = qz_ballot(segstart);
        const uint64_t Sle = S & (qz_below(lane) | (1ull << lane));
        const int ss = Sle ? qz_msb64(Sle) : -1;                /* my group's first lane in this wave; -1: it began in an earlier wave */
        const bool wave_open = (S & 1ull) == 0;                 /* the wave's first group continues one of an earlier wave */
        /* an open wave looks at its neighbour's elements every round (the carry): their keys and group starts are static */
        uint32_t kprev = 0; uint64_t Sprev = 0;
        if (wave_open) {                                         /* wave-uniform (wave 0 never is) */
            const uint32_t idx = ((wv - 1) << 6) + (uint32_t)lane;
            kprev = L->keyA[idx];
            const uint32_t ks = qz_shfl(kprev, lane - 1);
            const uint32_t km = lane ? ks : (idx ? L->keyA[idx - 1] : ~kprev);
            Sprev = qz_ballot(idx == 0 || (km >> 10) != (kprev >> 10));
        }
        /* the chain as the table held it at the window's start: head[h] (one gather per position, the same value for the
         * whole group) and up to three links, with zlib's validity rules (a first candidate may lie exactly MAX_DIST back, a
         * chained one must lie above the limit; at or below the window origin is NIL), and the 16 speculative bytes of each.
         * Kept packed: this kernel lives at its register limit.
         *   cp01 / cp23  c0 | c1 << 16, c2 | c3 << 16
         *   plv          four 5-bit lengths, then bits 20.. : c0 valid as first candidate, as chained one, c1, c2, c3 valid */
        uint32_t cp01 = 0, cp23 = 0, plv = 0;
        {
            const uint32_t lo = (p - base > QZK_MAXDIST) ? p - QZK_MAXDIST : base;     /* chained candidates must lie above it */
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            if (canh) {
                c0 = L->w.pre0[t];
                if (c0 > base && p - c0 <= QZK_MAXDIST) {
                    c1 = L->prev[c0 & 0x7fff];
                    if (c1 > lo) { c2 = L->prev[c1 & 0x7fff]; if (c2 > lo) c3 = L->prev[c2 & 0x7fff]; }
                }
            }
            const bool v0_head = canh && c0 > base && p - c0 <= QZK_MAXDIST, v0_chain = canh && c0 > lo;
            const bool v1 = v0_head && c1 > lo, v2 = v1 && c2 > lo, v3 = v2 && c3 > lo;
            uint32_t own[4], x[4];
            qzx_ld16(in32, p < 65536u ? p : 65536u, own);
            uint32_t pl0 = 0, pl1 = 0, pl2 = 0, pl3 = 0;
            if (v0_head) { qzx_ld16(in32, c0, x); pl0 = qzx_len16(own, x); }
            if (v1) { qzx_ld16(in32, c1, x); pl1 = qzx_len16(own, x); }
            if (v2) { qzx_ld16(in32, c2, x); pl2 = qzx_len16(own, x); }
            if (v3) { qzx_ld16(in32, c3, x); pl3 = qzx_len16(own, x); }
            cp01 = c0 | (c1 << 16); cp23 = c2 | (c3 << 16);
            plv = pl0 | (pl1 << 5) | (pl2 << 10) | (pl3 << 15) | (v0_head ? 1u << 20 : 0) | (v0_chain ? 1u << 21 : 0) |
                  (v1 ? 1u << 22 : 0) | (v2 ? 1u << 23 : 0) | (v3 ? 1u << 24 : 0);
        }

        QZX_DBG(" static done\n");
        QZX_T(2);
        /* ================= 3. rounds ================= */
        bool flag = canh && p != 0;                              /* round 0 assumes every position of the window inserted; position 0 is zlib's NIL */
        /* my in-window candidates (window offsets, newest first, 0xffff = none) as two packed pairs, their 16-byte lengths
         * (5 bits each) and their number (bits 20-22) */
        uint32_t lip01 = 0xffffffffu, lip23 = 0xffffffffu, llp = 0;
        uint32_t fin = QZX_NOPOS;                                /* my list's head | its length << 16 under the FINAL inserted set (the chain link) */
        uint32_t cache = 0;                                      /* position role: distance | full length << 16 of the extension that is known */
        uint32_t X = 0, adv = 0;                                 /* uniform: end of the final prefix / where the parse left the window */
        uint64_t Pmask = 0;                                      /* position role: this wave's parse points */
        uint32_t my_len = 0, my_dist = 0;                        /* position role: final match of position ws + tid */
        for (uint32_t round = 0;; round++) {
            /* ---- selection under `flag`: the four newest flagged members of my group before me ---- */
            const uint64_t F = qz_ballot(flag);
            uint32_t np01 = 0xffffffffu, np23 = 0xffffffffu, nn = 0;
            {
                /* carry: flagged members of my wave's FIRST group in earlier waves, newest first.  No exchange with those
                 * waves: the sorted keys are static and the inserted set of the last parse is in LDS, so this wave works
                 * out the flags of its neighbour's elements itself (one more wave back only when a whole wave belongs
                 * to the group and holds fewer than four inserted members - runs) */
                uint32_t cy0 = QZX_NOPOS, cy1 = QZX_NOPOS, cy2 = QZX_NOPOS, cy3 = QZX_NOPOS, ncy = 0;
                if (wave_open) {                                   /* wave-uniform */
                    for (int j = (int)wv - 1; j >= 0 && ncy < 4; j--) {
                        uint32_t kj = kprev; uint64_t Sj = Sprev;         /* the neighbour wave's keys and group starts are the window's */
                        if (j != (int)wv - 1) {
                            const uint32_t idx = ((uint32_t)j << 6) + (uint32_t)lane;
                            kj = L->keyA[idx];
                            const uint32_t kjs = qz_shfl(kj, lane - 1);
                            const uint32_t kjm = lane ? kjs : (idx ? L->keyA[idx - 1] : ~kj);
                            Sj = qz_ballot(idx == 0 || (kjm >> 10) != (kj >> 10));
                        }
                        const uint32_t tj = kj & 1023u, pj = ws + tj;
                        bool fj = pj + 3 <= n && pj != 0;
                        if (round > 0) {
                            uint32_t bit = (uint32_t)(L->insw[tj >> 6] >> (tj & 63u)) & 1u;
                            if ((tj & 63u) < 3) bit |= (L->spill[tj >> 6] >> (tj & 63u)) & 1u;
                            fj = fj && bit != 0;
                        }
                        uint64_t ml = qz_ballot(fj);
                        if (Sj) ml &= ~qz_below(qz_msb64(Sj));        /* the members of wave j's LAST group */
                        while (ml && ncy < 4) {
                            const int k = qz_msb64(ml); ml &= ~(1ull << k);
                            const uint32_t e = qz_readlane(tj, k);
                            if (ncy == 0) cy0 = e; else if (ncy == 1) cy1 = e; else if (ncy == 2) cy2 = e; else cy3 = e;
                            ncy++;
                        }
                        if (Sj) break;                                 /* the group began inside wave j */
                    }
                }
                uint64_t M = F & qz_below(lane);
                if (ss > 0) M &= ~qz_below(ss);                    /* lanes of my group only */
                int k0 = lane, k1 = lane, k2 = lane, k3 = lane;
                uint32_t cw = 0;
                if (M) { k0 = qz_msb64(M); M &= ~(1ull << k0); cw = 1; }
                if (M) { k1 = qz_msb64(M); M &= ~(1ull << k1); cw = 2; }
                if (M) { k2 = qz_msb64(M); M &= ~(1ull << k2); cw = 3; }
                if (M) { k3 = qz_msb64(M); cw = 4; }
                const uint32_t t0 = qz_shfl(t, k0), t1 = qz_shfl(t, k1), t2 = qz_shfl(t, k2), t3 = qz_shfl(t, k3);
                if (canh) {
                    uint32_t n0 = cw > 0 ? t0 : QZX_NOPOS, n1 = cw > 1 ? t1 : QZX_NOPOS, n2 = cw > 2 ? t2 : QZX_NOPOS, n3 = cw > 3 ? t3 : QZX_NOPOS;
                    nn = cw;
                    if (ss < 0 && nn < 4) {                       /* my group began before this wave: the carry follows */
                        uint32_t k = 0;
                        while (nn < 4 && k < ncy) {
                            const uint32_t e = k == 0 ? cy0 : k == 1 ? cy1 : k == 2 ? cy2 : cy3;
                            if (nn == 0) n0 = e; else if (nn == 1) n1 = e; else if (nn == 2) n2 = e; else n3 = e;
                            nn++; k++;
                        }
                    }
                    np01 = n0 | (n1 << 16); np23 = n2 | (n3 << 16);
                }
            }
            QZX_T(3); QZX_C(11, 1);
            /* ---- the check: did my list change against the one my match was computed with? ---- */
            const bool changed = np01 != lip01 || np23 != lip23;
            if (round > 0) {
                const bool pp = (L->ppw[t >> 6] >> (t & 63)) & 1ull;
                uint32_t *const box = &L->u[4 + (round & 1u)];            /* two mailboxes take turns: one barrier a round */
                if (changed && pp) atomicMin(box, t);
                if (tid == 0) L->u[4 + ((round + 1u) & 1u)] = QZX_NOPOS;
                qz_block_sync();
                const uint32_t xm = *box;
                fin = (np01 & 0xffffu) | (nn << 16);
                if (xm == QZX_NOPOS) { X = adv; break; }                            /* the window is final up to where its parse left it */
                if (round >= QZX_RMAX || 4 * xm >= 3 * lim) { X = xm; break; }      /* cut: the parse points before xm are final */
            }
            QZX_T(4);
            /* ---- matches for the lists that are new ---- */
            const bool redo = canh && (changed || round == 0);
            if (qz_ballot(redo) == 0) { /* nothing new in this wave */ }
            else if (redo) {
                const uint32_t maxlen = avail < 258 ? avail : 258, nice = avail < QZK_NICE ? avail : QZK_NICE;
                if (changed) {
                    uint32_t own[4], x[4];
                    qzx_ld16(in32, p, own);
                    uint32_t nl = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t nk = ((i < 2 ? np01 : np23) >> (16 * (i & 1))) & 0xffffu;
                        uint32_t lk = 0;
                        if (nk != QZX_NOPOS) {
                            if (nk == (lip01 & 0xffffu)) lk = llp & 31u;
                            else if (nk == (lip01 >> 16)) lk = (llp >> 5) & 31u;
                            else if (nk == (lip23 & 0xffffu)) lk = (llp >> 10) & 31u;
                            else if (nk == (lip23 >> 16)) lk = (llp >> 15) & 31u;
                            else { qzx_ld16(in32, ws + nk, x); lk = qzx_len16(own, x); }
                        }
                        nl |= lk << (5 * i);
                    }
                    lip01 = np01; lip23 = np23; llp = nl | (nn << 20);
                }
                const uint32_t nin = llp >> 20;
                uint32_t best = 2, bq = 0, cnt = 0; bool done = false;
#define QZX_TRY(q_, l_) do { uint32_t l__ = (l_); if (l__ > maxlen) l__ = maxlen; cnt++; if (l__ > best) { best = l__; bq = (q_); } if (l__ >= nice) done = true; } while (0)
                if (nin > 0) QZX_TRY(ws + (lip01 & 0xffffu), llp & 31u);
                if (nin > 1 && !done) QZX_TRY(ws + (lip01 >> 16), (llp >> 5) & 31u);
                if (nin > 2 && !done) QZX_TRY(ws + (lip23 & 0xffffu), (llp >> 10) & 31u);
                if (nin > 3 && !done) QZX_TRY(ws + (lip23 >> 16), (llp >> 15) & 31u);
                if (cnt < 4 && !done && ((plv >> (cnt == 0 ? 20 : 21)) & 1u)) {
                    QZX_TRY(cp01 & 0xffffu, plv & 31u);
                    if (cnt < 4 && !done && ((plv >> 22) & 1u)) {
                        QZX_TRY(cp01 >> 16, (plv >> 5) & 31u);
                        if (cnt < 4 && !done && ((plv >> 23) & 1u)) {
                            QZX_TRY(cp23 & 0xffffu, (plv >> 10) & 31u);
                            if (cnt < 4 && !done && ((plv >> 24) & 1u)) QZX_TRY(cp23 >> 16, (plv >> 15) & 31u);
                        }
                    }
                }
#undef QZX_TRY
                const uint32_t ml = best >= 3 ? best : 0;
                const bool capped = ml == QZX_CAP && maxlen > QZX_CAP;
                L->m.mlen[t] = (uint16_t)(ml | (capped ? 0x8000u : 0u));
                L->m.mdist[t] = (uint16_t)(ml ? p - bq : 0);
            }
            if (round == 0 && !canh) { L->m.mlen[t] = 0; L->m.mdist[t] = 0; }     /* fewer than three bytes ahead: a literal */
            qz_block_sync();

            QZX_DBG("  round %u matches done\n", round);
            if (round == 0) QZX_T(5); else QZX_T(13);
            /* ---- position order from here: thread = position ws + tid ---- */
            const uint32_t pp_ = ws + tid;
            const uint32_t avail_p = pp_ < n ? n - pp_ : 0;
            const uint32_t maxlen_p = avail_p < 258 ? avail_p : 258;
            uint32_t mlr = L->m.mlen[tid];
            const uint32_t mdr = L->m.mdist[tid];
            uint32_t len = mlr & 0x7fffu;
            {   /* matches that outgrew the 16 speculative bytes: consecutive positions with the same distance are one run - the
                 * run's first position in this wave goes on comparing, 16 bytes a step (far enough for every follower in the
                 * wave: 258 + 63), the others subtract their offset.  All inside the wave: no exchange, no barrier. */
                const bool capped = (mlr & 0x8000u) != 0;
                const bool need = capped && !((cache >> 16) && (cache & 0xffffu) == mdr);
                const uint32_t dprev = qz_shfl(mdr, lane - 1);
                const uint32_t nprev = qz_shfl(need ? 1u : 0u, lane - 1);
                const bool hd = need && (lane == 0 || !nprev || dprev != mdr);
                const uint64_t HD = qz_ballot(hd);
                uint32_t lx = QZX_CAP;
                if (HD) {                                                          /* wave-uniform */
                    if (hd) {
                        const uint32_t room = n - pp_;                             /* bytes from here to the chunk's end */
                        const uint32_t top = room < 258 + 64 ? room : 258 + 64;
                        while (lx < top) {
                            uint32_t a[4], b[4];
                            qzx_ld16(in32, pp_ + lx, a); qzx_ld16(in32, pp_ - mdr + lx, b);
                            const uint32_t l = qzx_len16(a, b);
                            lx += l;
                            if (l < QZX_CAP) break;
                        }
                        if (lx > top) lx = top;
                    }
                    const uint64_t hm = HD & (qz_below(lane) | (1ull << lane));
                    const int hl = hm ? qz_msb64(hm) : lane;                       /* my run's first lane (a follower's chain of equal distances leads to it) */
                    const uint32_t lh = qz_shfl(lx, hl);
                    if (need) {
                        uint32_t lf = lh - (uint32_t)(lane - hl);
                        if (lf > maxlen_p) lf = maxlen_p;
                        cache = mdr | (lf << 16);
                    }
                }
                if (capped) len = cache >> 16;
                else cache = 0;                                                    /* a different match: the cached extension no longer applies */
            }
            my_len = len; my_dist = mdr;
            QZX_DBG("  ext done\n");
            QZX_T(6);
            /* ---- parse: where does it leave this wave from each position, which positions does it touch ---- */
            const uint32_t wend = (wv + 1) << 6;
            uint32_t J = tid + (len ? len : 1u);
            uint64_t R = 1ull << lane;
#pragma unroll
            for (int s = 0; s < 6; s++) {
                const bool inside = J < wend && J < lim;
                const int tl = inside ? (int)(J & 63u) : lane;
                const uint32_t Jn = qz_shfl(J, tl);
                const uint64_t Rn = qzx_shfl64(R, tl);
                if (inside) { J = Jn; R |= Rn; }
            }
            L->w.exitp[tid] = (uint16_t)J;
            qz_block_sync();
            QZX_T(7);
            /* the sixteen exit tables chained from the window's first position (every wave walks the chain itself: uniform
             * LDS reads, no exchange; fetching all sixteen tables at once and hopping with readlane measured the same) */
            uint32_t cur = 0;
            while (cur < (wv << 6) && cur < lim) cur = qz_readfirstlane(L->w.exitp[cur]);
            uint64_t P = 0;
            if (cur < wend && cur < lim) {
                P = qzx_readlane64(R, (int)(cur & 63u));
                cur = qz_readfirstlane(L->w.exitp[cur]);
            }
            while (cur < lim) cur = qz_readfirstlane(L->w.exitp[cur]);
            adv = cur;                                                             /* first parse point at or behind lim */
            if ((wv << 6) < lim) { if (lim - (wv << 6) < 64) P &= qz_below((int)(lim - (wv << 6))); } else P = 0;
            Pmask = P;
            {   /* the inserted set this parse implies */
                const bool canh_p = avail_p >= 3;
                const uint64_t CANH = qz_ballot(canh_p);
                const uint64_t SH = qz_ballot(len >= 3 && len <= QZK_MAXINS && avail_p - len >= 3);
                const uint64_t S4 = qz_ballot(len == 4);
                const uint64_t ps = P & SH, p4 = ps & S4;
                const uint64_t I = (P & CANH) | (ps << 1) | (ps << 2) | (p4 << 3);
                const uint32_t sp = (uint32_t)(ps >> 63) | (uint32_t)(ps >> 62) | (uint32_t)(p4 >> 61);
                if (lane == 0) { L->insw[wv] = I; L->ppw[wv] = P; L->spill[wv + 1] = sp & 7u; if (wv == 0) L->spill[0] = 0; }
            }
            qz_block_sync();
            QZX_DBG("  parse done adv=%u\n", adv);
            QZX_T(8);
            /* ---- back to sorted order: is my element inserted? ---- */
            {
                const uint32_t wq = t >> 6, bq2 = t & 63u;
                uint32_t bit = (uint32_t)(L->insw[wq] >> bq2) & 1u;
                if (bq2 < 3) bit |= (L->spill[wq] >> bq2) & 1u;
                flag = canh && p != 0 && bit != 0;
            }
        }

        QZX_DBG(" rounds done X=%u\n", X);
        QZX_T(4);
        /* ================= 4. the final prefix [0, X): symbols, block marks, chains ================= */
        {
            const uint32_t pp_ = ws + tid;
            const uint32_t lo64 = wv << 6;
            uint64_t Pf = Pmask;
            if (X <= lo64) Pf = 0; else if (X - lo64 < 64) Pf &= qz_below((int)(X - lo64));
            const bool pp = (Pf >> lane) & 1ull;
            if (lane == 0) L->wtot[wv] = (uint32_t)qz_popc64(Pf);
            if (tid == 0) { L->u[2] = 0xffffffffu; L->u[3] = 0xffffffffu; }
            qz_block_sync();
            uint32_t before = 0, total = 0;
            for (uint32_t w = 0; w < QZX_NW; w++) { const uint32_t c = L->wtot[w]; if (w < wv) before += c; total += c; }
            const uint32_t idx = nsym + before + qzx_rank(Pf, lane);
            const uint32_t step = my_len ? my_len : 1u;
            if (pp) {
                olc[idx] = (uint8_t)(my_len ? my_len - 3 : (qzx_ld4(in32, pp_) & 0xffu));
                odist[idx] = (uint16_t)my_dist;
                if ((idx + 1) % QZK_LITBUF == 0) { L->u[2] = pp_ + step; L->u[3] = pp_; }       /* completes a block (at most one per window) */
            }
            /* chains (sorted order): an inserted position links to the newest inserted member of its group before it - the
             * first entry of its final list - or to what head[] held; the group's last inserted member becomes its head */
            const bool mine = flag && t < X;
            const uint64_t Ff = qz_ballot(mine);
            const uint32_t link = (fin >> 16) ? ws + (fin & 0xffffu) : (cp01 & 0xffffu);
            /* does my wave's last group go on, with an inserted member, in the waves behind me? */
            if (lane == 0) {
                const uint64_t first = S ? qz_below(qz_ctz64(S)) : ~0ull;          /* lanes of the wave's first group when it is a continuation */
                qzx_pub pb; pb.cnt = (wave_open && (Ff & first)) ? 1u : 0u; pb.cont = S == 0 ? 1u : 0u; pb.hlast = wave_open ? 1u : 0u;
                pb.e[0] = pb.e[1] = pb.e[2] = pb.e[3] = 0; pb.pad = 0;
                L->pub[wv] = pb;
            }
            qz_block_sync();
            bool later = false;                                                    /* uniform */
            for (uint32_t j = wv + 1; j < QZX_NW; j++) {
                const qzx_pub pb = L->pub[j];
                if (!pb.hlast) break;                                              /* wave j starts a new group */
                if (pb.cnt) { later = true; break; }
                if (!pb.cont) break;                                               /* the group ends inside wave j, nobody inserted */
            }
            if (mine) {
                L->prev[p & 0x7fff] = (uint16_t)link;
                uint64_t above = Ff & ~(qz_below(lane) | (1ull << lane));          /* inserted lanes above me ... */
                const uint64_t nextstart = S & ~(qz_below(lane) | (1ull << lane));
                if (nextstart) above &= qz_below(qz_ctz64(nextstart));             /* ... that are still in my group */
                const bool in_last_group = nextstart == 0;
                if (!above && !(in_last_group && later)) head[h] = (uint16_t)p;
            }
            const uint32_t nb = L->u[2], closer = L->u[3];
            if (nb != 0xffffffffu) {
                const uint32_t base_at = closer >= slide_thr ? (uint32_t)QZK_WSIZE : 0u;
                if (cur_bstart >= base_at) can_store |= 1u << nfull;
                nfull++;
                cur_bstart = nb;
                if (nfull < QZK_MAXBLK) mt->bstart[nfull] = nb;
            }
            nsym += total;
            ws += X;
            qzx_drain_stores();                                                    /* the next window's gathers see this window's heads */
            qz_block_sync();
            QZX_T(9);
        }
    }
    /* zlib's final loop top (lookahead == 0) may still slide before the last flush */
    {
        const uint32_t base_end = n >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST) ? (uint32_t)QZK_WSIZE : 0u;
        if (cur_bstart >= base_end) can_store |= 1u << nfull;
    }
    mt->nsym = nsym; mt->nfull = nfull; mt->can_store = can_store; mt->n = n;         /* uniform, every thread stores the same */
#ifndef QZ_SIM
    if (xprof && wv == 0) {
        const uint64_t t_ = __builtin_readcyclecounter() - L->xp[14];
        if (lane < 16) xprof[(size_t)chunk * 16 + (uint32_t)lane] = lane == 14 ? t_ : lane == 15 ? 0 : L->xp[lane];
    }
#endif
}

/* persistent workgroups, one per CU: each pulls chunk numbers; head[] of workgroup g at headtab + g * 65536 */
QZ_KERNEL_MAX(QZX_W) qzk_lz77_wide_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                                          uint8_t *sym_lc, uint16_t *sym_dist, qzk_lzmeta *meta, uint16_t *headtab,
                                          uint32_t *counter, const uint32_t *cdesc, uint64_t *xprof /* NULL, or 16 words per chunk */)
{
    QZ_LDS qzx_lds L;
    uint16_t *head = headtab + (size_t)blockIdx.x * 65536;
    for (;;) {
        if (threadIdx.x == 0) L.u[8] = atomicAdd(counter, 1u);
        qz_block_sync();
        const uint32_t chunk = L.u[8];
        qz_block_sync();
        if (chunk >= nchunks) break;
        const uint64_t coff = (uint64_t)chunk * chunk_sz;
        qzx_chunk(src, src_len, chunk_sz, chunk, sym_lc + coff, sym_dist + coff, meta, head, cdesc, &L, xprof);
    }
}

#endif
