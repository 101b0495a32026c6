/*
 * qzk_deflate_lz77_lane.h — K1b: zlib-exact greedy LZ77 parse with ONE CHUNK PER LANE, gfx950.
 *
 * Same output contract as qzk_lz77_kernel (qzk_deflate_lz77.h: symbol arrays + qzk_lzmeta,
 * consumed by K2) and the same place in the reference (zlib deflate_fast behind
 * qzDeflateSWCompress, src/qatzip_sw.c:178-231), opposite mapping.  The parse of a chunk
 * is a serial dependency chain, and a 2 GiB call is 32 768 independent chunks: here every
 * LANE runs zlib's loop on its own chunk with its own head[] / prev[] tables in HBM
 * (192 KiB per chunk), so a call keeps tens of thousands of chains in flight instead of two
 * per CU.  Each lane is latency-bound (a handful of dependent HBM/L2 round trips per
 * symbol); throughput comes from the lane count, which is why the host selects this kernel
 * only for calls with many chunks and keeps the LDS-resident wave kernel for small ones.
 *
 * The per-lane code is zlib's algorithm stated directly: rolling 16-bit hash of 3 bytes,
 * head/prev chains with NIL == 0, MAX_DIST = 32506 (head <=, chain <), blocks cut at 32767
 * symbols, the window slide at strstart >= 65274.  prev[] stores distances so only head[] is
 * rebased.  Because it IS zlib's loop with zlib's tables, it takes every row of zlib's
 * configuration_table: levels 1-3 are the greedy parse (deflate_fast: interiors inserted only
 * for matches <= max_insert), levels 4-9 the lazy one (deflate_slow: a match is held back one
 * byte and dropped for a literal if the next position matches longer; search shortened to a
 * quarter once the held match is `good`, skipped at `lazy`; 3-byte matches further than 4096
 * back are not taken; every position of an emitted match is inserted).  The product path uses
 * it for levels 2-9; level 1 has its own kernel (qzk_deflate_lz77.h).
 */
#ifndef QZK_DEFLATE_LZ77_LANE_H
#define QZK_DEFLATE_LZ77_LANE_H
#include "qzk_deflate_lz77.h"

typedef struct __attribute__((packed, aligned(1))) { uint64_t v; } qzk_u64u;

/* common prefix of in[a..] and in[b..] (b < a), at most maxlen, never reading past src_len */
QZ_DEV int qzk_lane_matchlen(const uint8_t *src, uint64_t src_len, uint64_t a, uint64_t b, int maxlen)
{
    int len = 0;
    while (len + 8 <= maxlen && a + len + 8 <= src_len) {
        const uint64_t x = ((const qzk_u64u *)(src + a + len))->v ^ ((const qzk_u64u *)(src + b + len))->v;
        if (x) return len + (__builtin_ctzll(x) >> 3);
        len += 8;
    }
    while (len < maxlen && a + len < src_len && src[a + len] == src[b + len]) len++;
    return len;
}

/* one row of zlib's configuration_table (deflate.c), plus which of its two parsers the row names */
typedef struct { int good, lazy, nice, chain, slow; } qzk_lvlcfg;

static inline qzk_lvlcfg qzk_level_cfg(int level)          /* level 1..9 */
{
    static const qzk_lvlcfg rows[10] = {{0, 0, 0, 0, 0}, {4, 4, 8, 4, 0}, {4, 5, 16, 8, 0}, {4, 6, 32, 32, 0}, {4, 4, 16, 16, 1},
                                        {8, 16, 32, 32, 1}, {8, 16, 128, 128, 1}, {8, 32, 128, 256, 1}, {32, 128, 258, 1024, 1},
                                        {32, 258, 258, 4096, 1}};
    return rows[level < 1 || level > 9 ? 1 : level];
}

QZ_KERNEL qzk_lz77_lane_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                               uint8_t *sym_lc, uint16_t *sym_dist, qzk_lzmeta *meta,
                               uint16_t *head_all /* zeroed by the host */, uint16_t *prev_all, qzk_lvlcfg cfg,
                               const uint32_t *cdesc)
{
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= nchunks) return;
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    uint8_t *olc = sym_lc + coff;
    uint16_t *odist = sym_dist + coff;
    qzk_lzmeta *mt = meta + chunk;
    uint16_t *head = head_all + (uint64_t)chunk * QZK_HSIZE;
    uint16_t *prev = prev_all + (uint64_t)chunk * QZK_WSIZE;
    const uint8_t *in = src + coff;

    uint32_t base = 0, fill = n < 65536u ? n : 65536u, avail_in = n - fill, pos = 0;
    uint32_t nsym = 0, nfull = 0, cur_bstart = 0, can_store = 0, inblock = 0;
    uint32_t match_len = 2, match_at = 0;               /* lazy parse: the match found at the previous position (chunk offset of its source) */
    bool held = false;                                  /* lazy parse: byte pos-1 waits for this position's verdict */
    mt->bstart[0] = 0;

    /* INSERT_STRING at chunk offset q: returns the previous head of its chain (window position, 0 == NIL) */
#define QZK_LANE_INSERT(q, hh) do { \
        const uint32_t pq_ = (q) - base; \
        const uint32_t h_ = (((uint32_t)(in[q] & 0xf) << 12) ^ ((uint32_t)in[(q) + 1] << 6) ^ in[(q) + 2]) & 0xffff; \
        (hh) = head[h_]; \
        prev[pq_ & (QZK_WSIZE - 1)] = (uint16_t)(((hh) != 0 && pq_ - (hh) <= 32767u) ? pq_ - (hh) : 0); \
        head[h_] = (uint16_t)pq_; } while (0)
    /* _tr_tally: the symbol, then the block-full rule; `nb` is where the next block's input starts */
#define QZK_LANE_TALLY(lc, d, full) do { olc[nsym] = (uint8_t)(lc); odist[nsym] = (uint16_t)(d); nsym++; \
        (full) = ++inblock == QZK_LITBUF; } while (0)
#define QZK_LANE_CLOSE(nb) do { if (cur_bstart >= base) can_store |= 1u << nfull; \
        nfull++; inblock = 0; cur_bstart = (nb); if (nfull < QZK_MAXBLK) mt->bstart[nfull] = (nb); } while (0)

    for (;;) {
        uint32_t look = fill - pos;
        if (look < QZK_MINLOOK) {                       /* zlib fill_window() */
            if (pos - base >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST)) {
                base += QZK_WSIZE;
                for (int i = 0; i < QZK_HSIZE / 2; i++) {
                    uint32_t v = ((uint32_t *)head)[i], lo = v & 0xffff, hi = v >> 16;
                    lo = lo >= QZK_WSIZE ? lo - QZK_WSIZE : 0;
                    hi = hi >= QZK_WSIZE ? hi - QZK_WSIZE : 0;
                    ((uint32_t *)head)[i] = lo | (hi << 16);
                }
            }
            if (avail_in) {
                uint32_t more = 65536u - (fill - base), rd = avail_in < more ? avail_in : more;
                fill += rd; avail_in -= rd;
            }
            look = fill - pos;
            if (look == 0) break;
        }
        const uint32_t p = pos - base;                  /* window position (zlib strstart) */
        uint32_t hash_head = 0;
        if (look >= 3) QZK_LANE_INSERT(pos, hash_head);

        const uint32_t prev_len = match_len, prev_at = match_at;    /* greedy levels: match_len stays 2 between positions */
        uint32_t mlen = 2;
        if (hash_head != 0 && prev_len < (uint32_t)(cfg.slow ? cfg.lazy : 3) && p - hash_head <= QZK_MAXDIST) {  /* longest_match */
            const int maxlen = look < 258 ? (int)look : 258, nice = look < (uint32_t)cfg.nice ? (int)look : cfg.nice;
            const int limit = p > QZK_MAXDIST ? (int)(p - QZK_MAXDIST) : 0;
            int best = (int)prev_len, cur = (int)hash_head, chain = prev_len >= (uint32_t)cfg.good ? cfg.chain >> 2 : cfg.chain;
            do {
                const int len = qzk_lane_matchlen(src, src_len, coff + pos, coff + base + (uint32_t)cur, maxlen);
                if (len > best) { best = len; match_at = base + (uint32_t)cur; if (len >= nice) break; }
                const int d = prev[cur & (QZK_WSIZE - 1)];
                cur = d ? cur - d : 0;
            } while (cur > limit && --chain != 0);
            mlen = (uint32_t)best <= look ? (uint32_t)best : look;
            if (cfg.slow && mlen == 3 && pos - match_at > 4096u) mlen = 2;     /* TOO_FAR */
        }
        bool full;
        if (!cfg.slow) {                                /* deflate_fast */
            if (mlen >= 3) {
                QZK_LANE_TALLY(mlen - 3, pos - match_at, full);
                if (mlen <= (uint32_t)cfg.lazy && look - mlen >= 3) {       /* insert the interiors of a short match */
                    for (uint32_t k = 1; k < mlen; k++) { uint32_t hh; QZK_LANE_INSERT(pos + k, hh); (void)hh; }
                }
                pos += mlen;
            } else {
                QZK_LANE_TALLY(in[pos], 0, full);
                pos++;
            }
            if (full) QZK_LANE_CLOSE(pos);
        } else if (prev_len >= 3 && mlen <= prev_len) { /* deflate_slow: the held match stands */
            QZK_LANE_TALLY(prev_len - 3, pos - 1 - prev_at, full);
            for (uint32_t k = 1; k + 1 < prev_len; k++)                     /* pos-1 and pos are in already */
                if (pos + k + 3 <= fill) { uint32_t hh; QZK_LANE_INSERT(pos + k, hh); (void)hh; }
            pos += prev_len - 1;
            held = false; match_len = 2;
            if (full) QZK_LANE_CLOSE(pos);
        } else {
            if (held) {                                 /* nothing held, or a longer match here: byte pos-1 is a literal */
                QZK_LANE_TALLY(in[pos - 1], 0, full);
                if (full) QZK_LANE_CLOSE(pos);          /* the block ends before the byte that is now held */
            }
            held = true; match_len = mlen;
            pos++;
        }
    }
    if (held) { bool full; QZK_LANE_TALLY(in[pos - 1], 0, full); (void)full; }     /* no block-full check here, as in zlib */
    if (cur_bstart >= base) can_store |= 1u << nfull;
    mt->nsym = nsym; mt->nfull = nfull; mt->can_store = can_store; mt->n = n;
#undef QZK_LANE_INSERT
#undef QZK_LANE_TALLY
#undef QZK_LANE_CLOSE
}

#endif
