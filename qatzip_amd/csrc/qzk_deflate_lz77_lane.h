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
 * head/prev chains with NIL == 0, MAX_DIST = 32506 (head <=, chain <), chain length 4,
 * nice length 8, interiors inserted only for matches <= 4, blocks cut at 32767 symbols, the
 * window slide at strstart >= 65274.  prev[] stores distances so only head[] is rebased.
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

QZ_KERNEL qzk_lz77_lane_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                               uint8_t *sym_lc, uint16_t *sym_dist, qzk_lzmeta *meta,
                               uint16_t *head_all /* zeroed by the host */, uint16_t *prev_all)
{
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= nchunks) return;
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = (uint32_t)((src_len - coff) < chunk_sz ? (src_len - coff) : chunk_sz);
    uint8_t *olc = sym_lc + coff;
    uint16_t *odist = sym_dist + coff;
    qzk_lzmeta *mt = meta + chunk;
    uint16_t *head = head_all + (uint64_t)chunk * QZK_HSIZE;
    uint16_t *prev = prev_all + (uint64_t)chunk * QZK_WSIZE;
    const uint8_t *in = src + coff;

    uint32_t base = 0, fill = n < 65536u ? n : 65536u, avail_in = n - fill, pos = 0;
    uint32_t nsym = 0, nfull = 0, cur_bstart = 0, can_store = 0, inblock = 0;
    mt->bstart[0] = 0;

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
        uint32_t hash_head = 0, mlen = 0, mdist = 0;
        if (look >= 3) {                                /* INSERT_STRING */
            const uint32_t h = (((uint32_t)(in[pos] & 0xf) << 12) ^ ((uint32_t)in[pos + 1] << 6) ^ in[pos + 2]) & 0xffff;
            hash_head = head[h];
            prev[p & (QZK_WSIZE - 1)] = (uint16_t)((hash_head != 0 && p - hash_head <= 32767u) ? p - hash_head : 0);
            head[h] = (uint16_t)p;
        }
        if (hash_head != 0 && p - hash_head <= QZK_MAXDIST) {       /* longest_match */
            const int maxlen = look < 258 ? (int)look : 258, nice = look < QZK_NICE ? (int)look : QZK_NICE;
            const int limit = p > QZK_MAXDIST ? (int)(p - QZK_MAXDIST) : 0;
            int best = 2, cur = (int)hash_head, chain = 4, mstart = 0;
            do {
                const int len = qzk_lane_matchlen(src, src_len, coff + pos, coff + base + (uint32_t)cur, maxlen);
                if (len > best) { best = len; mstart = cur; if (len >= nice) break; }
                const int d = prev[cur & (QZK_WSIZE - 1)];
                cur = d ? cur - d : 0;
            } while (cur > limit && --chain != 0);
            if (best >= 3) { mlen = (uint32_t)best; mdist = p - (uint32_t)mstart; }
        }
        /* tally */
        olc[nsym] = (uint8_t)(mlen ? mlen - 3 : in[pos]);
        odist[nsym] = (uint16_t)mdist;
        nsym++;
        const uint32_t step = mlen ? mlen : 1;
        if (mlen && mlen <= QZK_MAXINS && look - mlen >= 3) {       /* insert the interiors of a short match */
            for (uint32_t k = 1; k < mlen; k++) {
                const uint32_t q = pos + k, pq = q - base;
                const uint32_t h = (((uint32_t)(in[q] & 0xf) << 12) ^ ((uint32_t)in[q + 1] << 6) ^ in[q + 2]) & 0xffff;
                const uint32_t hh = head[h];
                prev[pq & (QZK_WSIZE - 1)] = (uint16_t)((hh != 0 && pq - hh <= 32767u) ? pq - hh : 0);
                head[h] = (uint16_t)pq;
            }
        }
        pos += step;
        if (++inblock == QZK_LITBUF) {                              /* _tr_tally: block full */
            if (cur_bstart >= base) can_store |= 1u << nfull;
            nfull++; inblock = 0;
            cur_bstart = pos;
            if (nfull < QZK_MAXBLK) mt->bstart[nfull] = pos;
        }
    }
    if (cur_bstart >= base) can_store |= 1u << nfull;
    mt->nsym = nsym; mt->nfull = nfull; mt->can_store = can_store; mt->n = n;
}

#endif
