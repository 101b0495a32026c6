/*
 * qzk_lz4.h — K4 (LZ4 block compress, lz4 1.9.3 LZ4_compress_fast acceleration 1 semantics),
 * K5 (LZ4 frame decode) and XXH32 (K6) for gfx950, one frame per wave.
 *
 * Replaces LZ4F_compressFrame / LZ4F_decompress on the reference's software path
 * (src/qatzip_sw.c:443-533) for frames of at most one 64 KB block — the
 * "64 KB blocks + xxhash32" configuration of BASELINE.json.  A call above 64 KB is
 * what liblz4 makes of it: ONE frame whose blocks are linked (qzk_lz4c_linked_kernel:
 * LZ4_compress_fast_continue over the blocks, one parse state for the frame - a
 * serial chain, so one wave per call).
 * CPU restatement: oracle/qzo_lz4.c.
 *
 * Compress: the parse is greedy and serial (every probed position is inserted in
 * the table; a hit ends the streak), so the wave speculates a WINDOW of the next 64
 * probe positions at once — hash, table lookup (u32 positions in LDS, inserted with
 * atomicMax so that order inside the window does not matter), 4-byte compare — and
 * takes the first hit.  Only lanes whose hash collides with an earlier lane of the
 * same window need an exact replay, done with readlanes.  Match extension and the
 * literal copies are wave-parallel.
 */
#ifndef QZK_LZ4_H
#define QZK_LZ4_H
#include "qzk_common.h"
#include "qzk_lz_batch.h"

#define QZK_LZ4_MINMATCH 4
#define QZK_LZ4_MFLIMIT 12
#define QZK_LZ4_LASTLIT 5
#define QZK_LZ4_HASHSZ 8192
#define QZK_LZ4_MAXBLK 65536
#ifndef QZK_LZ4_W0
#define QZK_LZ4_W0 16u             /* probes in the first window of a search streak */
#endif

#define QZK_XP1 2654435761u
#define QZK_XP2 2246822519u
#define QZK_XP3 3266489917u
#define QZK_XP4 668265263u
#define QZK_XP5 374761393u
QZ_DEV uint32_t qzk_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* XXH32(p[0..n), seed 0) by one wave: lanes 0-3 carry the four stripe accumulators */
QZ_DEV uint32_t qzk_wave_xxh32(const uint8_t *p, uint32_t n, int lane)
{
    uint32_t h, pos = 0;
    if (n >= 16) {
        uint32_t v = lane == 0 ? QZK_XP1 + QZK_XP2 : lane == 1 ? QZK_XP2 : lane == 2 ? 0u : 0u - QZK_XP1;
        const uint32_t stripes = n >> 4;
        if (lane < 4) {
            /* the accumulator chain is serial, the loads are not: eight stripes' words are asked for before the first is used */
            uint32_t s = 0;
            for (; s + 8 <= stripes; s += 8) {
                uint32_t w[8];
#pragma unroll
                for (int k = 0; k < 8; k++) w[k] = qz_ld32(p + 16 * (s + (uint32_t)k) + 4 * lane);
#pragma unroll
                for (int k = 0; k < 8; k++) v = qzk_rotl(v + w[k] * QZK_XP2, 13) * QZK_XP1;
            }
            for (; s < stripes; s++) v = qzk_rotl(v + qz_ld32(p + 16 * s + 4 * lane) * QZK_XP2, 13) * QZK_XP1;
        }
        uint32_t v0 = qz_readlane(v, 0), v1 = qz_readlane(v, 1), v2 = qz_readlane(v, 2), v3 = qz_readlane(v, 3);
        h = qzk_rotl(v0, 1) + qzk_rotl(v1, 7) + qzk_rotl(v2, 12) + qzk_rotl(v3, 18);
        pos = stripes << 4;
    } else h = QZK_XP5;
    h += n;
    while (pos + 4 <= n) { h = qzk_rotl(h + qz_ld32(p + pos) * QZK_XP3, 17) * QZK_XP4; pos += 4; }
    while (pos < n) { h = qzk_rotl(h + p[pos] * QZK_XP5, 11) * QZK_XP1; pos++; }
    h ^= h >> 15; h *= QZK_XP2; h ^= h >> 13; h *= QZK_XP3; h ^= h >> 16;
    return h;
}

/* XXH32 of fewer than sixteen bytes (a frame header's descriptor): no stripes, wave-uniform */
QZ_DEV uint32_t qzk_xxh32_small(const uint8_t *p, uint32_t n)
{
    uint32_t h = QZK_XP5 + n, pos = 0;
    while (pos + 4 <= n) { h = qzk_rotl(h + qz_ld32(p + pos) * QZK_XP3, 17) * QZK_XP4; pos += 4; }
    while (pos < n) { h = qzk_rotl(h + p[pos] * QZK_XP5, 11) * QZK_XP1; pos++; }
    h ^= h >> 15; h *= QZK_XP2; h ^= h >> 13; h *= QZK_XP3; h ^= h >> 16;
    return h;
}

/* The same hash for a frame's content (tens of KB): the four accumulator chains stay serial, but their words arrive as
 * whole 16-byte stripes - every lane fetches two of the next 128 while lanes 0-3 work through the 128 before them out of
 * `stage` (2 KiB of LDS, 16-byte aligned), instead of four lanes fetching dwords 16 bytes apart.  Round 5: the decoder got
 * ~10x faster (qzk_lz4d_kernel below) and four lanes' loads had become a fifth of a frame. */
QZ_DEV uint32_t qzk_wave_xxh32_staged(const uint8_t *p, uint32_t n, uint8_t *stage, int lane)
{
    if (n < 16 * 128) return qzk_wave_xxh32(p, n, lane);
    const uint32_t stripes = n >> 4;
    uint32_t v = lane == 0 ? QZK_XP1 + QZK_XP2 : lane == 1 ? QZK_XP2 : lane == 2 ? 0u : 0u - QZK_XP1;
    /* a stripe = two 8-byte pieces (the content may sit at any byte address) */
    uint64_t a0 = qzk_rb_ld64(p + 16 * (uint32_t)lane), a1 = qzk_rb_ld64(p + 16 * (uint32_t)lane + 8);
    uint64_t b0 = qzk_rb_ld64(p + 16 * (64 + (uint32_t)lane)), b1 = qzk_rb_ld64(p + 16 * (64 + (uint32_t)lane) + 8);
    for (uint32_t s0 = 0; s0 < stripes; s0 += 128) {
        const uint32_t cnt = stripes - s0 < 128 ? stripes - s0 : 128;
        qz_lds_sync();                                              /* the stripes before have been worked through */
        ((uint64_t *)stage)[2 * (uint32_t)lane] = a0; ((uint64_t *)stage)[2 * (uint32_t)lane + 1] = a1;
        ((uint64_t *)stage)[2 * (64 + (uint32_t)lane)] = b0; ((uint64_t *)stage)[2 * (64 + (uint32_t)lane) + 1] = b1;
        const uint32_t s1 = s0 + 128;                               /* the next 128: asked for now */
        if (s1 + (uint32_t)lane < stripes) { a0 = qzk_rb_ld64(p + 16 * (s1 + (uint32_t)lane)); a1 = qzk_rb_ld64(p + 16 * (s1 + (uint32_t)lane) + 8); }
        if (s1 + 64 + (uint32_t)lane < stripes) { b0 = qzk_rb_ld64(p + 16 * (s1 + 64 + (uint32_t)lane)); b1 = qzk_rb_ld64(p + 16 * (s1 + 64 + (uint32_t)lane) + 8); }
        qz_lds_sync();
        if (lane < 4) {
            const uint32_t *w = (const uint32_t *)stage + lane;
            uint32_t k = 0;
            for (; k + 8 <= cnt; k += 8) {                          /* eight words asked for before the first is used */
                const uint32_t x0 = w[4 * k], x1 = w[4 * k + 4], x2 = w[4 * k + 8], x3 = w[4 * k + 12], x4 = w[4 * k + 16],
                               x5 = w[4 * k + 20], x6 = w[4 * k + 24], x7 = w[4 * k + 28];
#define QZK_XR(x_) v = qzk_rotl(v + (x_) * QZK_XP2, 13) * QZK_XP1
                QZK_XR(x0); QZK_XR(x1); QZK_XR(x2); QZK_XR(x3); QZK_XR(x4); QZK_XR(x5); QZK_XR(x6); QZK_XR(x7);
#undef QZK_XR
            }
            for (; k < cnt; k++) v = qzk_rotl(v + w[4 * k] * QZK_XP2, 13) * QZK_XP1;
        }
    }
    const uint32_t v0 = qz_readlane(v, 0), v1 = qz_readlane(v, 1), v2 = qz_readlane(v, 2), v3 = qz_readlane(v, 3);
    uint32_t h = qzk_rotl(v0, 1) + qzk_rotl(v1, 7) + qzk_rotl(v2, 12) + qzk_rotl(v3, 18);
    uint32_t pos = stripes << 4;
    h += n;
    while (pos + 4 <= n) { h = qzk_rotl(h + qz_ld32(p + pos) * QZK_XP3, 17) * QZK_XP4; pos += 4; }
    while (pos < n) { h = qzk_rotl(h + p[pos] * QZK_XP5, 11) * QZK_XP1; pos++; }
    h ^= h >> 15; h *= QZK_XP2; h ^= h >> 13; h *= QZK_XP3; h ^= h >> 16;
    return h;
}

QZ_DEV void qzk_wave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, int lane)
{
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
}

/* common prefix of a[] and b[] (b < a, may overlap), at most maxlen bytes */
QZ_DEV uint32_t qzk_lz4_count(const uint8_t *a, const uint8_t *b, uint32_t maxlen, int lane)
{
    for (uint32_t off = 0; off < maxlen; off += 256) {
        uint32_t o = off + 4 * (uint32_t)lane, x = 0;
        bool act = o < maxlen;
        if (act) {
            if (o + 4 <= maxlen) x = qz_ld32(a + o) ^ qz_ld32(b + o);
            else for (uint32_t k = 0; o + k < maxlen; k++) x |= (uint32_t)(a[o + k] ^ b[o + k]) << (8 * k);
        }
        uint64_t mm = qz_ballot(act && x != 0);
        if (mm) {
            int f = qz_ctz64(mm);
            uint32_t xf = qz_readlane(x, f);
            return off + 4 * (uint32_t)f + ((uint32_t)qz_ctz32(xf) >> 3);
        }
    }
    return maxlen;
}

#define QZK_LZ4HASH(v) (((v) * 2654435761u) >> 19)

/* lz4's LZ4_hash5 of the 64-bit little-endian builds: the five bytes at p, 12 bits (the table of the linked mode) */
QZ_DEV uint32_t qzk_lz4_hash5(const uint8_t *p)
{
    const uint64_t seq = (uint64_t)qz_ld32(p) | (uint64_t)p[4] << 32;
    return (uint32_t)(((seq << 24) * 889523592379ull) >> 52);
}

/* LZ4 block compress of in[bs..bs+n) into out (capacity cap); returns size or 0 when it does not fit.
 * LINKED = false: an independent block (bs = 0; LZ4_compress_fast, 13-bit hash of 4 bytes; table: QZK_LZ4_HASHSZ u16 -
 * positions of a 64 KB block, lz4's own byU16 - so that eight of these waves fit a CU's LDS instead of four;
 * in LDS, zeroed here).  LINKED = true: one block of a linked frame that starts at in[0] (LZ4_compress_fast_continue:
 * the frame's table - 4096 u32, 12-bit hash of 5 bytes - comes in and goes out with everything earlier blocks and this
 * one inserted, also when this block does not fit; candidates up to 65535 bytes back, across block borders). */
/* GTAB: the (16-bit, independent-block) table lives in device memory instead of LDS - 16 KiB per wave that then do not
 * decide how many waves a CU holds (qzk_lz4c_pull_kernel).  A lookup is served by the L2 (`sc1`: this CU's L1 may hold the
 * line from before one of this wave's own stores, which write through and do not refresh it); a wave's loads follow its
 * stores to the L2 in issue order. */
typedef uint32_t qzk_lz4_u32x4 __attribute__((vector_size(16)));
template <bool GTAB, typename TAB>
QZ_DEV uint32_t qzk_lz4_tld(const TAB *table, uint32_t h)
{
#ifndef QZ_SIM
    if (GTAB) {
        uint32_t v;
        asm volatile("global_load_ushort %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(table + h) : "memory");
        return v;
    }
#endif
    return table[h];
}

template <bool LINKED, typename TAB, bool GTAB = false>
/* Syncs: every exchange between the lanes here goes through LDS (slot table, LDS hash tables: a wave's DS operations
 * execute in order) or through the device-memory table, whose stores and `sc1` loads reach the L2 in issue order - so the
 * lanes only need the compiler kept from reordering (qz_lds_sync), not qz_wave_sync()'s workgroup fence: that one waits
 * (s_waitcnt vmcnt(0)) for every store still on its way, table inserts and output bytes included, a full round trip at
 * each of the ten syncs of a sequence (round 5). */
QZ_DEV uint32_t qzk_lz4_block_t(const uint8_t *in, uint32_t bs, uint32_t n, uint8_t *out, uint32_t cap, TAB *table,
                                uint32_t *slot, int lane)
{
#define QZK_LZ4H(pos, v4) (LINKED ? qzk_lz4_hash5(in + (pos)) : QZK_LZ4HASH(v4))
#define QZK_LZ4NEAR(cand, cur) (!LINKED || (cand) + 65535u >= (cur))
    if (!LINKED) {
        if (GTAB) { for (int i = lane; i < (int)(QZK_LZ4_HASHSZ * sizeof(TAB) / 16); i += 64) { qzk_lz4_u32x4 z = {0, 0, 0, 0}; ((qzk_lz4_u32x4 *)table)[i] = z; } }
        else { for (int i = lane; i < QZK_LZ4_HASHSZ; i += 64) table[i] = 0; }
    }
    qz_lds_sync();
    const uint32_t be = bs + n;
    uint32_t op = 0, anchor = bs, ip;
    const int32_t mfl1 = (int32_t)be - QZK_LZ4_MFLIMIT + 1;     /* mflimitPlusOne */
    const uint32_t matchlimit = be - QZK_LZ4_LASTLIT;
    bool ended = false;
    if (n >= QZK_LZ4_MFLIMIT + 1) {
        /* first byte: the block's first position goes into the table unsearched (an independent block's is 0 - already
         * there), the search starts behind it */
        if (LINKED) { if (lane == 0) table[qzk_lz4_hash5(in + bs)] = (TAB)bs; qz_lds_sync(); }
        ip = bs + 1;
        for (;;) {
            /* ---------------- search streak from ip ---------------- */
            uint32_t j0 = 0, mpos = 0, mcand = 0; bool found = false;
            for (;;) {
                /* probe j of a streak sits at ip + S(j): lz4 advances by step = 1 for the first probe and then by
                 * (searchMatchNb++ >> 6) with searchMatchNb starting at 64, i.e. step_j = (63 + j) >> 6 for j >= 1 */
                /* a streak's first window is QZK_LZ4_W0 probes wide: on data that compresses the hit is usually among
                 * them, and every probe costs a table lookup and a candidate compare at an address of its own (the CU's
                 * address path is what this kernel runs into); streaks that go on take the whole wave */
                const uint32_t W = j0 == 0 ? QZK_LZ4_W0 : 64u;
                const bool inw = (uint32_t)lane < W;
                const uint32_t j = j0 + (uint32_t)lane, t = j ? j - 1 : 0, b = t >> 6, r = t & 63;
                const uint32_t f = ip + (j ? 1 + 32 * b * (b + 1) + (b + 1) * r : 0), step = j ? (63 + j) >> 6 : 1;
                const bool can = (int32_t)(f + step) <= mfl1;        /* else this probe is the `goto _last_literals` */
                const bool live = inw && can;
                uint32_t v = 0, h = 0, cand = 0;
                if (live) { v = qz_ld32(in + f); h = QZK_LZ4H(f, v); cand = qzk_lz4_tld<GTAB>(table, h); }
                const uint32_t key = h & 1023;
                if (live) slot[key] = 64;
                qz_lds_sync();
                if (live) atomicMin(&slot[key], (uint32_t)lane);
                qz_lds_sync();
                const bool suspect = live && slot[key] != (uint32_t)lane;
                bool hit = false;
                if (live && !suspect) hit = QZK_LZ4NEAR(cand, f) && qz_ld32(in + cand) == v;
                const uint64_t DEAD = qz_ballot(inw && !can);
                uint64_t HIT = qz_ballot(hit);
                const uint64_t SUS = qz_ballot(suspect);
                const int first_dead = DEAD ? qz_ctz64(DEAD) : (int)W;
                int bound = HIT ? qz_ctz64(HIT) : (int)W;
                if (bound > first_dead) bound = first_dead;
                /* replay colliding lanes that come before the first clean hit */
                uint64_t todo = SUS & qz_below(bound);
                int fl = bound; uint32_t fcand = 0;
                while (todo) {
                    int s = qz_ctz64(todo);
                    todo &= todo - 1;
                    uint32_t hs = qz_readlane(h, s), vs = qz_readlane(v, s);
                    uint64_t same = qz_ballot(live && h == hs) & qz_below(s);
                    uint32_t cs, cv; bool near = true;
                    if (same) { int m = qz_msb64(same); cs = qz_readlane(f, m); cv = qz_readlane(v, m); }
                    else { cs = qz_readlane(cand, s); cv = qz_ld32(in + cs); near = QZK_LZ4NEAR(cs, qz_readlane(f, s)); }
                    if (near && cv == vs) { fl = s; fcand = cs; break; }
                }
                if (fl == bound && bound < first_dead && HIT) fcand = qz_readlane(cand, bound);
                const bool got = fl < first_dead && (fl < bound || (HIT && bound < (int)W && fl == bound));
                /* insert every probed position up to and including the hit (or the whole window) */
                const int last_ins = got ? fl : (first_dead < (int)W ? first_dead - 1 : (int)W - 1);
                if (sizeof(TAB) == 4) {
                    if (live && lane <= last_ins) atomicMax((uint32_t *)&table[h], f);
                    qz_lds_sync();
                } else {
                    /* 16-bit entries have no LDS atomic: the last probe of a hash is elected through the slot table - per
                     * key the highest (rest of the hash, lane) wins, its hash is served, the other hashes on that key go
                     * round again (eight hashes share a key, two of them in one window is already rare) */
                    bool pend = live && lane <= last_ins;
                    const uint32_t val = (((h >> 10) << 6) | (uint32_t)lane) + 1;
                    while (qz_ballot(pend)) {
                        if (pend) slot[key] = 0;
                        qz_lds_sync();
                        if (pend) atomicMax(&slot[key], val);
                        qz_lds_sync();
                        const uint32_t w = pend ? slot[key] : 0;
                        if (pend && w == val) table[h] = (TAB)f;
                        if (pend && ((w - 1) >> 6) == (h >> 10)) pend = false;      /* my hash was the one served */
                        qz_lds_sync();
                    }
                }
                if (got) { found = true; mpos = qz_readlane(f, fl); mcand = fcand; break; }
                if (first_dead < (int)W) break;                      /* ran into the end: last literals */
                j0 += W;
            }
            if (!found) break;
            /* ---------------- catch up + sequence(s) ---------------- */
            uint32_t mip = mpos, match = mcand;
            {   /* backward extension */
                uint32_t maxb = mip - anchor < match ? mip - anchor : match;
                uint32_t back = 0;
                while (back < maxb) {
                    uint32_t i = back + 1 + (uint32_t)lane;
                    bool act = i <= maxb;
                    bool ne = act && in[mip - i] != in[match - i];
                    uint64_t mm = qz_ballot(ne), am = qz_ballot(act);
                    if (mm) { back += (uint32_t)qz_ctz64(mm); break; }
                    back += (uint32_t)qz_popc64(am);
                }
                mip -= back; match -= back;
            }
            uint32_t lit = mip - anchor, token_at = op++;
            if (op + lit + (2 + 1 + QZK_LZ4_LASTLIT) + lit / 255 > cap) return 0;
            uint32_t tok;
            if (lit >= 15) {
                uint32_t len = lit - 15; tok = 15u << 4;
                for (; len >= 255; len -= 255) { if (lane == 0) out[op] = 255; op++; }
                if (lane == 0) out[op] = (uint8_t)len;
                op++;
            } else tok = lit << 4;
            qzk_wave_copy(out + op, in + anchor, lit, lane); op += lit;
            for (;;) {          /* _next_match */
                if (lane == 0) { out[op] = (uint8_t)(mip - match); out[op + 1] = (uint8_t)((mip - match) >> 8); }
                op += 2;
                uint32_t mc = qzk_lz4_count(in + mip + 4, in + match + 4, matchlimit - (mip + 4), lane);
                mip += mc + 4;
                if (op + (1 + QZK_LZ4_LASTLIT) + (mc + 240) / 255 > cap) return 0;
                if (mc >= 15) {
                    tok += 15; mc -= 15;
                    for (; mc >= 255; mc -= 255) { if (lane == 0) out[op] = 255; op++; }
                    if (lane == 0) out[op] = (uint8_t)mc;
                    op++;
                } else tok += mc;
                if (lane == 0) out[token_at] = (uint8_t)tok;
                anchor = mip;
                if ((int32_t)mip >= mfl1) { ended = true; break; }
                /* fill table with ip-2, then test the next position right away */
                uint32_t v2 = qz_ld32(in + mip - 2), v0 = qz_ld32(in + mip);
                uint32_t h2 = QZK_LZ4H(mip - 2, v2), h0 = QZK_LZ4H(mip, v0);
                if (lane == 0) table[h2] = (TAB)(mip - 2);
                qz_lds_sync();
                uint32_t mi = qzk_lz4_tld<GTAB>(table, h0);
                qz_lds_sync();
                if (lane == 0) table[h0] = (TAB)mip;
                qz_lds_sync();
                if (QZK_LZ4NEAR(mi, mip) && qz_ld32(in + mi) == v0) { token_at = op++; tok = 0; match = mi; continue; }
                break;
            }
            if (ended) break;
            ip = mip + 1;
        }
    }
    /* last literals */
    {
        uint32_t lr = be - anchor;
        if (op + lr + 1 + (lr + 255 - 15) / 255 > cap) return 0;
        if (lr >= 15) {
            uint32_t acc = lr - 15;
            if (lane == 0) out[op] = 15u << 4;
            op++;
            for (; acc >= 255; acc -= 255) { if (lane == 0) out[op] = 255; op++; }
            if (lane == 0) out[op] = (uint8_t)acc;
            op++;
        } else { if (lane == 0) out[op] = (uint8_t)(lr << 4); op++; }
        qzk_wave_copy(out + op, in + anchor, lr, lane); op += lr;
    }
    return op;
#undef QZK_LZ4H
#undef QZK_LZ4NEAR
}
template <bool GTAB>
QZ_DEV uint32_t qzk_lz4_block(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, uint16_t *table, uint32_t *slot, int lane)
{ return qzk_lz4_block_t<false, uint16_t, GTAB>(in, 0, n, out, cap, table, slot, lane); }

/* ------------------------------------------------------------------ K4, round 5: the block compressor of the pull kernel
 * The same parse (lz4 1.9.3 LZ4_compress_fast, 16-bit positions, 13-bit hash), rebuilt around the number of dependent
 * memory round trips per sequence - ten in the form above (profiles/r4_lz4_counters.txt: 49.8 ms per GiB, the waves waiting
 * 80 % of their cycles at 32 waves per CU: nothing but fewer round trips makes a wave faster), three here:
 *   - a table entry carries what the compare needs: {epoch, position, the four bytes at that position} in 8 bytes.  The
 *     lookup's answer says hit or miss; the gather of 16-64 candidate words (a line each) is gone.  The epoch - a counter
 *     the wave keeps, one per frame - replaces the clearing of the table: an entry of another frame reads as lz4's
 *     zero-initialised slot (position 0, with the block's first four bytes);
 *   - the input around the parse point lives in a register window F (lane i = the dword at fb + 4i, 256 bytes): the probes'
 *     own words, the literals, the catch-up and the forward count on the input's side, and the two words the test behind a
 *     match needs all come out of it through cross-lane reads; it is re-based (one load) every couple of hundred bytes;
 *   - on a hit ONE load brings the candidate's neighbourhood (C: 16 bytes before it, 240 behind) - catch-up and count are
 *     compares of F against C in registers, and only a match longer than that goes back to memory;
 *   - the block is put together in LDS and leaves as whole 16-byte rows (tokens, length bytes and offsets were single-byte
 *     stores of lane 0).
 * A streak that goes beyond its first sixteen probes (data that does not compress) reads its probes' words from memory as
 * before; literal runs above 128 bytes are copied memory to memory. */
static_assert(QZK_LZ4_W0 <= 64u && QZK_LZ4_W0 + 4u + 128u + 16u <= 256u, "the first probe window and the literals behind it fit the register window");
#define QZK_L4C_OST 832u            /* staging: a flush leaves < 256 + 16 bytes, a sequence adds at most 1 + 2 + 128 + 2 + 258 */
#define QZK_L4C_LITMAX 128u
typedef struct { uint8_t *st; uint8_t *out; uint32_t oph, op, adj, hd; } qzk_l4o;
#define QZK_L4O_IDX(O_, p_) ((uint32_t)(p_) + (O_)->adj)
QZ_DEV void qzk_l4o_init(qzk_l4o *O, uint8_t *st, uint8_t *out)
{
    O->st = st; O->out = out; O->oph = (uint32_t)((uintptr_t)out & 15); O->op = 0; O->adj = O->oph; O->hd = O->oph;
}
/* whole rows out, the ragged one to the front (as qzk_lz_batch.h) */
QZ_DEV void qzk_l4o_rows(qzk_l4o *O, int lane)
{
    const uint32_t total = QZK_L4O_IDX(O, O->op), R = total >> 4;
    if (!R) return;
    qz_lds_sync();
    uint8_t *const g0 = O->out - (int32_t)O->adj;                          /* 16-byte aligned (adj = phase - 16 * rows gone) */
    uint32_t first = 0;
    if (O->hd) { if ((uint32_t)lane >= O->hd && lane < 16) g0[lane] = O->st[lane]; first = 1; O->hd = 0; }
    for (uint32_t rw = first + (uint32_t)lane; rw < R; rw += 64) *(qzk_rb_u32x4 *)(g0 + 16 * rw) = *(const qzk_rb_u32x4 *)(O->st + 16 * rw);
    qz_lds_sync();
    if ((total & 15) && lane < 4) ((uint32_t *)O->st)[lane] = ((const uint32_t *)O->st)[4 * R + (uint32_t)lane];
    O->adj -= 16 * R;
    qz_lds_sync();
}
/* everything out (before a direct copy, at the end) */
QZ_DEV void qzk_l4o_all(qzk_l4o *O, int lane)
{
    qzk_l4o_rows(O, lane);
    const uint32_t sh = QZK_L4O_IDX(O, O->op);                     /* < 16 */
    qz_lds_sync();
    if ((uint32_t)lane >= O->hd && (uint32_t)lane < sh) (O->out - (int32_t)O->adj)[lane] = O->st[lane];
    O->hd = sh;
    qz_lds_sync();                                                  /* the staging is written again from here on */
}
QZ_DEV void qzk_l4o_byte(qzk_l4o *O, uint32_t b, int lane) { if (lane == 0) O->st[QZK_L4O_IDX(O, O->op)] = (uint8_t)b; O->op++; }
/* a length that did not fit its nibble: len - 15 in bytes of 255 */
QZ_DEV void qzk_l4o_len(qzk_l4o *O, uint32_t len, int lane)
{
    const uint32_t nff = len / 255u;
    uint8_t *d = O->st + QZK_L4O_IDX(O, O->op);
    for (uint32_t i = (uint32_t)lane; i < nff; i += 64) d[i] = 255;
    if (lane == 0) d[nff] = (uint8_t)(len - 255u * nff);
    O->op += nff + 1;
}

/* the register window: 256 bytes of the block from fb on, a dword per lane; bytes behind the block's end read as zero */
QZ_DEV uint32_t qzk_f_load(const uint8_t *in, uint32_t n, uint32_t fb, int lane)
{
    const uint32_t o = fb + 4u * (uint32_t)lane;
    if (o + 4 <= n) return qz_ld32(in + o);
    uint32_t v = 0;
    for (uint32_t t = 0; t < 4; t++) if (o + t < n) v |= (uint32_t)in[o + t] << (8 * t);
    return v;
}
#ifdef QZ_SIM
QZ_DEV uint32_t qzk_alignb(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * sh)); }
#else
QZ_DEV uint32_t qzk_alignb(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }     /* one v_alignbyte_b32 (sh = 0..3) */
#endif
/* the dword at byte r of the window, r of the lane's own (r + 4 <= 256; every lane takes part) */
QZ_DEV uint32_t qzk_f_x32(uint32_t F, uint32_t r)
{
    const uint32_t k = (r >> 2) & 63u;
    const uint32_t lo = qz_shfl(F, (int)k), hi = qz_shfl(F, (int)(k < 63u ? k + 1 : 63u));
    return qzk_alignb(hi, lo, r & 3u);
}
QZ_DEV uint32_t qzk_f_u32(uint32_t F, uint32_t r)                  /* the same for a wave-uniform r */
{
    const uint32_t k = r >> 2;
    const uint32_t lo = qz_readlane(F, (int)k), hi = qz_readlane(F, (int)(k < 63u ? k + 1 : 63u));
    return qzk_alignb(hi, lo, r & 3u);
}

QZ_DEV uint64_t qzk_lz4_tld64(const uint64_t *p)
{
#ifndef QZ_SIM
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
#else
    return *p;
#endif
}
#define QZK_L4E(ep_, pos_, v_) ((uint64_t)(ep_) << 48 | (uint64_t)(pos_) << 32 | (uint64_t)(v_))

/* LZ4 block compress of in[0..n) into out (capacity cap) through `table` (QZK_LZ4_HASHSZ entries of 8 bytes in device
 * memory, this wave's, never cleared per block: `epoch` != 0 is this block's); returns the size, 0 when it does not fit */
QZ_DEV uint32_t qzk_lz4_block_f(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, uint64_t *table, uint32_t epoch,
                                uint32_t *slot, uint8_t *stage, int lane)
{
    qzk_l4o O; qzk_l4o_init(&O, stage, out);
    const uint32_t be = n;
    uint32_t anchor = 0, ip;
    const int32_t mfl1 = (int32_t)be - QZK_LZ4_MFLIMIT + 1;     /* mflimitPlusOne */
    const uint32_t matchlimit = be - QZK_LZ4_LASTLIT;
    bool ended = false;
    if (n >= QZK_LZ4_MFLIMIT + 1) {
        const uint32_t in0 = qz_uniform(qz_ld32(in));               /* what lz4's zero-initialised table points at (wave-uniform: a scalar register) */
        uint32_t fb = 0, F = qzk_f_load(in, n, 0, lane);
#define QZK_L4DEC(e_, cand_, cv_) do { const uint32_t hi_ = (uint32_t)((e_) >> 32); const bool mine_ = (hi_ >> 16) == epoch; \
            cand_ = mine_ ? hi_ & 0xffffu : 0u; cv_ = mine_ ? (uint32_t)(e_) : in0; } while (0)
        ip = 1;
        for (;;) {
            /* ---------------- search streak from ip ---------------- */
            {   /* the window covers the literals so far (when they are few enough to go through it) and the first probes */
                const uint32_t lo = ip - anchor <= QZK_L4C_LITMAX ? anchor : (ip >= 16 ? ip - 16 : 0);
                if (fb > lo || ip + QZK_LZ4_W0 + 4 > fb + 256) { fb = lo; F = qzk_f_load(in, n, fb, lane); }   /* (the first window's probes and their four bytes) */
            }
            uint32_t j0 = 0, mpos = 0, mcand = 0; bool found = false;
            for (;;) {
                const uint32_t W = j0 == 0 ? QZK_LZ4_W0 : 64u;
                const bool inw = (uint32_t)lane < W;
                uint32_t f, step;
                if (j0 == 0) { f = ip + (uint32_t)lane; step = 1; }  /* the first 64 probes of a streak are one byte apart */
                else {
                    const uint32_t j = j0 + (uint32_t)lane, t = j - 1, b = t >> 6, r = t & 63;
                    f = ip + 1 + 32 * b * (b + 1) + (b + 1) * r; step = (63 + j) >> 6;
                }
                const bool can = (int32_t)(f + step) <= mfl1;        /* else this probe is the `goto _last_literals` */
                const bool live = inw && can;
                uint32_t v = 0, h = 0, cand = 0, cv = 0;
                if (j0 == 0) { const uint32_t x = qzk_f_x32(F, live ? f - fb : 0u); v = live ? x : 0u; }
                else if (live) v = qz_ld32(in + f);
                if (live) { h = QZK_LZ4HASH(v); const uint64_t e = qzk_lz4_tld64(table + h); QZK_L4DEC(e, cand, cv); }
                const uint32_t key = h & 1023;
                if (live) slot[key] = 64;
                qz_lds_sync();
                if (live) atomicMin(&slot[key], (uint32_t)lane);
                qz_lds_sync();
                const bool suspect = live && slot[key] != (uint32_t)lane;
                const bool hit = live && !suspect && cv == v;
                const uint64_t DEAD = qz_ballot(inw && !can);
                uint64_t HIT = qz_ballot(hit);
                const uint64_t SUS = qz_ballot(suspect);
                const int first_dead = DEAD ? qz_ctz64(DEAD) : (int)W;
                int bound = HIT ? qz_ctz64(HIT) : (int)W;
                if (bound > first_dead) bound = first_dead;
                /* replay colliding lanes that come before the first clean hit */
                uint64_t todo = SUS & qz_below(bound);
                int fl = bound; uint32_t fcand = 0;
                while (todo) {
                    int s = qz_ctz64(todo);
                    todo &= todo - 1;
                    uint32_t hs = qz_readlane(h, s), vs = qz_readlane(v, s);
                    uint64_t same = qz_ballot(live && h == hs) & qz_below(s);
                    uint32_t cs, cw;
                    if (same) { int m = qz_msb64(same); cs = qz_readlane(f, m); cw = qz_readlane(v, m); }
                    else { cs = qz_readlane(cand, s); cw = qz_readlane(cv, s); }
                    if (cw == vs) { fl = s; fcand = cs; break; }
                }
                if (fl == bound && bound < first_dead && HIT) fcand = qz_readlane(cand, bound);
                const bool got = fl < first_dead && (fl < bound || (HIT && bound < (int)W && fl == bound));
                /* insert every probed position up to and including the hit (or the whole window): per key the highest
                 * (rest of the hash, lane) wins, its hash is served, the other hashes on that key go round again */
                const int last_ins = got ? fl : (first_dead < (int)W ? first_dead - 1 : (int)W - 1);
                if (SUS == 0) {
                    /* no two lanes of the window share a key, let alone a hash: every lane is its hash's last writer */
                    if (live && lane <= last_ins) table[h] = QZK_L4E(epoch, f, v);
                    qz_lds_sync();                                  /* (the next window's lookups come behind these stores) */
                } else {
                    bool pend = live && lane <= last_ins;
                    const uint32_t val = (((h >> 10) << 6) | (uint32_t)lane) + 1;
                    while (qz_ballot(pend)) {
                        if (pend) slot[key] = 0;
                        qz_lds_sync();
                        if (pend) atomicMax(&slot[key], val);
                        qz_lds_sync();
                        const uint32_t w = pend ? slot[key] : 0;
                        if (pend && w == val) table[h] = QZK_L4E(epoch, f, v);
                        if (pend && ((w - 1) >> 6) == (h >> 10)) pend = false;      /* my hash was the one served */
                        qz_lds_sync();
                    }
                }
                if (got) { found = true; mpos = qz_readlane(f, fl); mcand = fcand; break; }
                if (first_dead < (int)W) break;                      /* ran into the end: last literals */
                j0 += W;
            }
            if (!found) break;
            /* ---------------- catch up + sequence(s) ---------------- */
            uint32_t mip = mpos, match = mcand;
            uint32_t cb = match >= 16 ? match - 16 : 0;
            uint32_t C = qzk_f_load(in, n, cb, lane);               /* the candidate's neighbourhood: the sequence's one load */
            const uint32_t maxb = mip - anchor < match ? mip - anchor : match;
            if (maxb != 0) {   /* backward extension: lane i compares the bytes i + 1 back, sixteen at a time out of the two windows */
                const uint32_t fhas = mip >= fb && mip <= fb + 256u ? mip - fb : 0u;      /* bytes before mip that F holds */
                const uint32_t reach = fhas < 16u ? fhas : 16u;                          /* (C holds sixteen, or all there is) */
                const uint32_t lim = maxb < reach ? maxb : reach;
                const uint32_t ro = mip - 1 - (uint32_t)lane - fb, rc = match - 1 - (uint32_t)lane - cb;
                const bool act = (uint32_t)lane < lim;
                const uint32_t ow = qz_shfl(F, act ? (int)(ro >> 2) : 0), cw = qz_shfl(C, act ? (int)(rc >> 2) : 0);
                const bool eq = act && ((ow >> (8 * (ro & 3))) & 0xff) == ((cw >> (8 * (rc & 3))) & 0xff);
                const uint64_t NE = ~qz_ballot(eq);
                uint32_t back = NE ? (uint32_t)qz_ctz64(NE) : 64u;
                if (back > lim) back = lim;
                if (back == lim && lim < maxb) {
                    /* everything at hand agreed and there may be more: the rest from memory (rare) */
                    while (back < maxb) {
                        uint32_t i = back + 1 + (uint32_t)lane;
                        bool a2 = i <= maxb;
                        bool ne = a2 && in[mip - i] != in[match - i];
                        uint64_t mm = qz_ballot(ne), am = qz_ballot(a2);
                        if (mm) { back += (uint32_t)qz_ctz64(mm); break; }
                        back += (uint32_t)qz_popc64(am);
                    }
                }
                mip -= back; match -= back;
            }
            const uint32_t lit = mip - anchor;
            if (QZK_L4O_IDX(&O, O.op) >= 256) qzk_l4o_rows(&O, lane);
            uint32_t token_at = O.op; bool tok_staged = true;
            if (O.op + 1 + lit + (2 + 1 + QZK_LZ4_LASTLIT) + lit / 255 > cap) return 0;
            qzk_l4o_byte(&O, 0, lane);
            uint32_t tok;
            if (lit >= 15) { tok = 15u << 4; qzk_l4o_len(&O, lit - 15, lane); } else tok = lit << 4;
            if (lit != 0 && lit <= QZK_L4C_LITMAX && anchor >= fb && mip <= fb + 256) {
                /* literals out of the window (lane i holds the window's bytes 4i .. 4i + 3) */
                const uint32_t a0 = anchor - fb;
                uint8_t *d = O.st + QZK_L4O_IDX(&O, O.op);
                for (uint32_t k0 = 0; k0 < lit; k0 += 64) {         /* literal k = byte a0 + k of the window: lane k fetches it from its owner */
                    const uint32_t k = k0 + (uint32_t)lane, r = a0 + k;
                    const uint32_t w = qz_shfl(F, (int)((r >> 2) & 63u));
                    if (k < lit) d[k] = (uint8_t)(w >> (8 * (r & 3u)));
                }
                O.op += lit;
            } else if (lit != 0) {
                qzk_l4o_all(&O, lane);                              /* token and length bytes first, then the run memory to memory */
                tok_staged = false;
                uint8_t *d = O.out + O.op; const uint8_t *sp = in + anchor;
                for (uint32_t i = 8 * (uint32_t)lane; i + 8 <= lit; i += 512) qzk_rb_st64(d + i, qzk_rb_ld64(sp + i));
                { const uint32_t t8 = lit & ~7u; if ((uint32_t)lane < (lit & 7u)) d[t8 + (uint32_t)lane] = sp[t8 + (uint32_t)lane]; }
                O.op += lit;
                O.adj = ((O.oph + O.op) & 15u) - O.op; O.hd = (O.oph + O.op) & 15u;
            }
            for (;;) {          /* _next_match */
                if (lane == 0) { uint8_t *d = O.st + QZK_L4O_IDX(&O, O.op); d[0] = (uint8_t)(mip - match); d[1] = (uint8_t)((mip - match) >> 8); }
                O.op += 2;
                /* forward count: the input's side from F, the candidate's from C, a dword per lane; what they do not hold
                 * (a match longer than ~230 bytes, a window that ends early) from memory */
                const uint32_t maxlen = matchlimit - (mip + 4);
                uint32_t mc;
                {
                    const uint32_t a0 = mip + 4 - fb, b0 = match + 4 - cb;             /* byte offsets in F and C (may lie outside) */
                    const bool fin = mip + 4 >= fb && a0 < 256u, cin = match + 4 >= cb && b0 < 256u;
                    const uint32_t fa = fin ? (256u - a0) >> 2 : 0, ca = cin ? (256u - b0) >> 2 : 0;    /* whole dwords at hand */
                    uint32_t nd = fa < ca ? fa : ca;                                    /* dwords both hold */
                    if (nd > (maxlen + 3) >> 2) nd = (maxlen + 3) >> 2;
                    const uint32_t ra = fin ? a0 + 4u * (uint32_t)lane : 0u, rb = cin ? b0 + 4u * (uint32_t)lane : 0u;
                    const bool act = (uint32_t)lane < nd;
                    const uint32_t xa = qzk_f_x32(F, act ? ra : 0u), xb = qzk_f_x32(C, act ? rb : 0u);
                    uint32_t x = act ? xa ^ xb : 0u;
                    const uint32_t left = maxlen - 4u * (uint32_t)lane;                 /* bytes of my dword that count */
                    if (act && left < 4u) x &= (1u << (8 * left)) - 1u;
                    const uint64_t mm = qz_ballot(x != 0);
                    if (mm) {
                        const int fl = qz_ctz64(mm);
                        const uint32_t xf = qz_readlane(x, fl);
                        mc = 4u * (uint32_t)fl + ((uint32_t)qz_ctz32(xf) >> 3);
                    } else {
                        const uint32_t covered = 4u * nd < maxlen ? 4u * nd : maxlen;
                        mc = covered;
                        if (covered < maxlen) mc += qzk_lz4_count(in + mip + 4 + covered, in + match + 4 + covered, maxlen - covered, lane);
                    }
                }
                mip += mc + 4;
                if (O.op + (1 + QZK_LZ4_LASTLIT) + (mc + 240) / 255 > cap) return 0;
                if (mc >= 15) { tok += 15; qzk_l4o_len(&O, mc - 15, lane); } else tok += mc;
                if (lane == 0) { if (tok_staged) O.st[QZK_L4O_IDX(&O, token_at)] = (uint8_t)tok; else O.out[token_at] = (uint8_t)tok; }
                anchor = mip;
                if ((int32_t)mip >= mfl1) { ended = true; break; }
                /* fill table with ip-2, then test the next position right away; the window moves up if it has to */
                if (fb + 2 > mip || mip + 1 + QZK_LZ4_W0 + 4 > fb + 256) { fb = mip >= 16 ? mip - 16 : 0; F = qzk_f_load(in, n, fb, lane); }
                const uint32_t v2 = qzk_f_u32(F, mip - 2 - fb), v0 = qzk_f_u32(F, mip - fb);
                const uint32_t h2 = QZK_LZ4HASH(v2), h0 = QZK_LZ4HASH(v0);
                if (lane == 0) table[h2] = QZK_L4E(epoch, mip - 2, v2);
                qz_lds_sync();
                const uint64_t e0 = qzk_lz4_tld64(table + h0);
                qz_lds_sync();
                if (lane == 0) table[h0] = QZK_L4E(epoch, mip, v0);
                qz_lds_sync();
                uint32_t mi, mv; QZK_L4DEC(e0, mi, mv);
                if (mv == v0) {
                    if (QZK_L4O_IDX(&O, O.op) >= 256) qzk_l4o_rows(&O, lane);
                    token_at = O.op; tok_staged = true; tok = 0; match = mi;
                    qzk_l4o_byte(&O, 0, lane);
                    cb = match >= 16 ? match - 16 : 0; C = qzk_f_load(in, n, cb, lane);
                    continue;
                }
                break;
            }
            if (ended) break;
            ip = mip + 1;
        }
#undef QZK_L4DEC
    }
    /* last literals */
    {
        const uint32_t lr = be - anchor;
        if (O.op + lr + 1 + (lr + 255 - 15) / 255 > cap) return 0;
        if (QZK_L4O_IDX(&O, O.op) >= 256) qzk_l4o_rows(&O, lane);
        if (lr >= 15) { qzk_l4o_byte(&O, 15u << 4, lane); qzk_l4o_len(&O, lr - 15, lane); } else qzk_l4o_byte(&O, lr << 4, lane);
        qzk_l4o_all(&O, lane);
        uint8_t *d = O.out + O.op; const uint8_t *sp = in + anchor;
        for (uint32_t i = 8 * (uint32_t)lane; i + 8 <= lr; i += 512) qzk_rb_st64(d + i, qzk_rb_ld64(sp + i));
        { const uint32_t t8 = lr & ~7u; if ((uint32_t)lane < (lr & 7u)) d[t8 + (uint32_t)lane] = sp[t8 + (uint32_t)lane]; }
        O.op += lr;
    }
    return O.op;
}

/* K4: one LZ4 frame (<= 64 KB of content, one independent block) per wave, written to its slot:
 * LZ4F_compressFrame with {contentChecksum, contentSize, autoFlush, level < 3}. */
/* hw_hdr: the header the reference's HARDWARE path puts in front of a chunk's frame (qzLZ4HeaderGen, src/qatzip_lz4.c:104-132):
 * FLG 0x4C - version 1, blocks NOT marked independent, content size always present, content checksum - instead of
 * liblz4's 0x6C (or 0x64 for an empty call); everything behind the header checksum byte is the same frame */
template <bool GTAB>
QZ_DEV void qzk_lz4c_frame(const uint8_t *src, uint64_t src_len, uint32_t frame_sz, uint32_t fr, uint8_t *slots, uint32_t stride,
                           uint32_t *out_len, uint32_t hw_hdr, uint16_t *table, uint32_t *slot, int lane)
{
    const uint64_t off = (uint64_t)fr * frame_sz;
    const uint32_t n = (uint32_t)((src_len - off) < frame_sz ? (src_len - off) : frame_sz);
    const uint8_t *in = src + off;
    uint8_t *o = slots + (uint64_t)fr * stride;
    uint32_t pos = 0;
    /* frame header: magic, FLG (v1 | independent | [content size] | content checksum), BD 64 KB */
    if (lane == 0) {
        o[0] = 0x04; o[1] = 0x22; o[2] = 0x4d; o[3] = 0x18;
        o[4] = hw_hdr ? (uint8_t)0x4C : (uint8_t)((1u << 6) | (1u << 5) | (n ? 1u << 3 : 0) | (1u << 2));
        o[5] = 4u << 4;
        if (n || hw_hdr) { o[6] = (uint8_t)n; o[7] = (uint8_t)(n >> 8); o[8] = (uint8_t)(n >> 16); o[9] = (uint8_t)(n >> 24); o[10] = o[11] = o[12] = o[13] = 0; }
    }
    qz_wave_sync();
    pos = (n || hw_hdr) ? 14 : 6;
    {
        uint32_t hc = qzk_wave_xxh32(o + 4, pos - 4, lane);
        if (lane == 0) o[pos] = (uint8_t)(hc >> 8);
        pos++;
    }
    if (n) {
        uint32_t c = qzk_lz4_block<GTAB>(in, n, o + pos + 4, n - 1, table, slot, lane);
        uint32_t bh = c ? c : (n | 0x80000000u);
        if (c == 0) { qzk_wave_copy(o + pos + 4, in, n, lane); c = n; }
        if (lane == 0) { o[pos] = (uint8_t)bh; o[pos + 1] = (uint8_t)(bh >> 8); o[pos + 2] = (uint8_t)(bh >> 16); o[pos + 3] = (uint8_t)(bh >> 24); }
        pos += 4 + c;
    }
    uint32_t xx = qzk_wave_xxh32(in, n, lane);
    if (lane == 0) {
        o[pos] = o[pos + 1] = o[pos + 2] = o[pos + 3] = 0;
        o[pos + 4] = (uint8_t)xx; o[pos + 5] = (uint8_t)(xx >> 8); o[pos + 6] = (uint8_t)(xx >> 16); o[pos + 7] = (uint8_t)(xx >> 24);
    }
    out_len[fr] = pos + 8;          /* wave-uniform: every lane stores the same word */
}

QZ_KERNEL_MAX(64) qzk_lz4c_kernel(const uint8_t *src, uint64_t src_len, uint32_t frame_sz, uint32_t nframes,
                          uint8_t *slots, uint32_t stride, uint32_t *out_len, uint32_t hw_hdr)
{
    QZ_LDS uint16_t table[QZK_LZ4_HASHSZ];
    QZ_LDS uint32_t slot[1024];
    const uint32_t fr = blockIdx.x;
    if (fr >= nframes) return;
    qzk_lz4c_frame<false>(src, src_len, frame_sz, fr, slots, stride, out_len, hw_hdr, table, slot, qz_lane());
}

/* K4 for calls of many frames: the same frames by PERSISTENT single-wave workgroups that pull frame numbers, with the hash
 * table of the wave in device memory (tables + wave * QZK_LZ4_HASHSZ) - only the 4 KiB slot table is left in LDS, so a CU
 * holds as many of these waves as it has wave slots instead of the eight that 20 KiB each allow.  The compressor is one
 * latency chain per frame (probes' words, table, candidates' words, extension, count - global round trips all of them;
 * profiles/r3_lz4_ring_experiment.txt: its rate follows the waves in flight, not the latency of one), so the extra round
 * trip to the L2 per lookup is paid back several times by the waves that now fit. */
#define QZK_L4C_TABW QZK_LZ4_HASHSZ          /* 8-byte entries per wave: 64 KiB of device memory */
#ifndef QZK_L4C_OCC
#define QZK_L4C_OCC 7            /* waves per SIMD the register budget is cut for (measured, ms per GiB: 6 - 80 VGPRs, no scratch - 44.3; 7 - 3 spilled - 42.1; 8 - 11 spilled - 41.6..42.6) */
#endif
QZ_KERNEL_OCC(64, QZK_L4C_OCC) qzk_lz4c_pull_kernel(const uint8_t *src, uint64_t src_len, uint32_t frame_sz, uint32_t nframes,
                          uint8_t *slots, uint32_t stride, uint32_t *out_len, uint32_t hw_hdr, uint64_t *tables, uint32_t *epochs, uint32_t *counter)
{
    QZ_LDS uint32_t slot[1024];
    QZ_LDS __attribute__((aligned(16))) uint8_t stage[QZK_L4C_OST];
    const int lane = qz_lane();
    uint64_t *const table = tables + (size_t)blockIdx.x * QZK_L4C_TABW;
    /* the wave's epoch lives on between launches (the table is only ever cleared when the 16 bits run out): cleared memory
     * and epoch 0 are what the host hands over */
    uint32_t ep = epochs[blockIdx.x];
    for (;;) {
        uint32_t fr = atomicAdd(counter, lane == 0 ? 1u : 0u);       /* every lane takes part, lane 0 adds (see qzk_lz77_pull_kernel) */
        fr = qz_readfirstlane(fr);
        if (fr >= nframes) break;
        if (++ep > 0xffffu) {
            for (uint32_t i = (uint32_t)lane; i < QZK_L4C_TABW; i += 64) table[i] = 0;
            ep = 1;
        }
        const uint64_t off = (uint64_t)fr * frame_sz;
        const uint32_t n = (uint32_t)((src_len - off) < frame_sz ? (src_len - off) : frame_sz);
        const uint8_t *in = src + off;
        uint8_t *o = slots + (uint64_t)fr * stride;
        /* frame header: magic, FLG (v1 | independent | [content size] | content checksum), BD 64 KB (qzk_lz4c_frame) */
        if (lane == 0) {
            o[0] = 0x04; o[1] = 0x22; o[2] = 0x4d; o[3] = 0x18;
            o[4] = hw_hdr ? (uint8_t)0x4C : (uint8_t)((1u << 6) | (1u << 5) | (n ? 1u << 3 : 0) | (1u << 2));
            o[5] = 4u << 4;
            if (n || hw_hdr) { o[6] = (uint8_t)n; o[7] = (uint8_t)(n >> 8); o[8] = (uint8_t)(n >> 16); o[9] = (uint8_t)(n >> 24); o[10] = o[11] = o[12] = o[13] = 0; }
        }
        qz_wave_sync();
        uint32_t pos = (n || hw_hdr) ? 14 : 6;
        {
            const uint32_t hc = qzk_xxh32_small(o + 4, pos - 4);        /* 2 or 10 bytes */
            if (lane == 0) o[pos] = (uint8_t)(hc >> 8);
            pos++;
        }
        if (n) {
            uint32_t c = qzk_lz4_block_f(in, n, o + pos + 4, n - 1, table, ep, slot, stage, lane);
            const uint32_t bh = c ? c : (n | 0x80000000u);
            if (c == 0) { qzk_wave_copy(o + pos + 4, in, n, lane); c = n; }
            if (lane == 0) { o[pos] = (uint8_t)bh; o[pos + 1] = (uint8_t)(bh >> 8); o[pos + 2] = (uint8_t)(bh >> 16); o[pos + 3] = (uint8_t)(bh >> 24); }
            pos += 4 + c;
        }
        qz_lds_sync();
        const uint32_t xx = qzk_wave_xxh32_staged(in, n, (uint8_t *)slot, lane);
        if (lane == 0) {
            o[pos] = o[pos + 1] = o[pos + 2] = o[pos + 3] = 0;
            o[pos + 4] = (uint8_t)xx; o[pos + 5] = (uint8_t)(xx >> 8); o[pos + 6] = (uint8_t)(xx >> 16); o[pos + 7] = (uint8_t)(xx >> 24);
        }
        out_len[fr] = pos + 8;          /* wave-uniform: every lane stores the same word */
        qz_wave_sync();
    }
    {   /* (the address is made again here, from scalar registers: kept alive across the loop it cost two VGPRs their place) */
        uint32_t *e2 = epochs;
#ifndef QZ_SIM
        asm volatile("" : "+s"(e2));
#endif
        if (lane == 0) e2[blockIdx.x] = ep;
    }
}

/* K4 for one call above 64 KB: the frame LZ4F_compressFrame writes for it (src/qatzip_sw.c:451-456) - FLG 0x4C (blocks
 * linked, content size, content checksum), then per 64 KB: block header + block (or the bytes themselves when the block
 * does not shrink), end mark, XXH32 of the content.  The blocks share one parse state, so this is one wave's serial work
 * from the first byte to the last; it is what keeps a QZ_LZ4 session's bytes equal to the software path's for every call
 * size, not a fast path (calls of at most 64 KB, and the 64 KB-frame bench configuration, go through qzk_lz4c_kernel). */
QZ_DEV uint32_t qzk_lz4c_linked_frame(const uint8_t *src, uint32_t n, uint8_t *out, uint32_t *table, uint32_t *slot, int lane)
{
    for (int i = lane; i < 4096; i += 64) table[i] = 0;
    if (lane == 0) {
        out[0] = 0x04; out[1] = 0x22; out[2] = 0x4d; out[3] = 0x18;
        out[4] = (uint8_t)((1u << 6) | (1u << 3) | (1u << 2));
        out[5] = 4u << 4;
        out[6] = (uint8_t)n; out[7] = (uint8_t)(n >> 8); out[8] = (uint8_t)(n >> 16); out[9] = (uint8_t)(n >> 24);
        out[10] = out[11] = out[12] = out[13] = 0;
    }
    qz_wave_sync();
    uint32_t pos = 14;
    {
        uint32_t hc = qzk_wave_xxh32(out + 4, pos - 4, lane);
        if (lane == 0) out[pos] = (uint8_t)(hc >> 8);
        pos++;
    }
    for (uint32_t bs = 0; bs < n; bs += QZK_LZ4_MAXBLK) {
        const uint32_t bn = n - bs < QZK_LZ4_MAXBLK ? n - bs : QZK_LZ4_MAXBLK;
        uint32_t c = qzk_lz4_block_t<true, uint32_t>(src, bs, bn, out + pos + 4, bn - 1, table, slot, lane);
        const uint32_t bh = c ? c : (bn | 0x80000000u);
        if (c == 0) { qz_wave_sync(); qzk_wave_copy(out + pos + 4, src + bs, bn, lane); c = bn; }
        if (lane == 0) { out[pos] = (uint8_t)bh; out[pos + 1] = (uint8_t)(bh >> 8); out[pos + 2] = (uint8_t)(bh >> 16); out[pos + 3] = (uint8_t)(bh >> 24); }
        pos += 4 + c;
        qz_wave_sync();
    }
    const uint32_t xx = qzk_wave_xxh32(src, n, lane);
    if (lane == 0) {
        out[pos] = out[pos + 1] = out[pos + 2] = out[pos + 3] = 0;
        out[pos + 4] = (uint8_t)xx; out[pos + 5] = (uint8_t)(xx >> 8); out[pos + 6] = (uint8_t)(xx >> 16); out[pos + 7] = (uint8_t)(xx >> 24);
    }
    return pos + 8;
}
QZ_KERNEL_MAX(64) qzk_lz4c_linked_kernel(const uint8_t *src, uint32_t n, uint8_t *out, uint32_t *out_len)
{
    QZ_LDS uint32_t table[4096];
    QZ_LDS uint32_t slot[1024];
    const int lane = qz_lane();
    const uint32_t len = qzk_lz4c_linked_frame(src, n, out, table, slot, lane);
    if (lane == 0) *out_len = len;
}
/* the hardware path's framing with a hw_buff_sz above 64 KB (src/qatzip_lz4.c:104-143): every chunk of `chunk` bytes is
 * such a frame of its own - one wave per chunk, all chunks of the call in one launch, frame k into slot k (the scan and the
 * gather of the one-block frames put them back to back).  The call's last chunk may be 64 KB or less: that one is a
 * one-block frame and not this kernel's (nfr counts the chunks above 64 KB only). */
QZ_KERNEL_MAX(64) qzk_lz4c_linked_many_kernel(const uint8_t *src, uint64_t total, uint32_t chunk, uint32_t nfr, uint8_t *slots,
                                              uint32_t stride, uint32_t *lens)
{
    QZ_LDS uint32_t table[4096];
    QZ_LDS uint32_t slot[1024];
    const int lane = qz_lane();
    const uint32_t k = blockIdx.x;
    if (k >= nfr) return;
    const uint64_t off = (uint64_t)k * chunk;
    const uint32_t n = total - off < chunk ? (uint32_t)(total - off) : chunk;
    const uint32_t len = qzk_lz4c_linked_frame(src + off, n, slots + (uint64_t)k * stride, table, slot, lane);
    if (lane == 0) lens[k] = len;
}

/* ------------------------------------------------------------------ K5: frame decode */
typedef struct { uint64_t in_off; uint64_t out_off; uint32_t in_len; uint32_t out_cap; } qzk_lz4seg;
typedef struct { int32_t status; uint32_t in_used; uint32_t out_len; uint32_t pad; } qzk_lz4res;
#define QZK_LZ4_OK 0
#define QZK_LZ4_EDATA (-1)
#define QZK_LZ4_EOUT (-2)
#define QZK_LZ4_EIN (-3)

/* K5, round 5: the frame's blocks through the batch engine of qzk_lz_batch.h.  An LZ4 block is a serial token stream - a
 * sequence's place depends on the literal runs before it - but what a sequence that BEGINS at a given byte would be depends
 * on a handful of bytes only.  So the walk is speculated a window at a time: lane i takes the block as if a sequence began
 * at s + i (token, literal run, offset, match length: two unaligned LDS reads out of a 1 KiB ring of the stream, refilled 512
 * bytes at a time with the next refill already on its way), and a short chase - v_readlane of the `next` field from s on, a
 * few scalar instructions a hop - picks the lanes that really are sequence starts.  Their records go into a queue in LDS in
 * stream order; whenever it holds 64 (or the block is through) the wave takes as many as fit the engine's window and literal
 * staging, one per lane, and resolves them in parallel: literals of the batch as one span of aligned rows into LDS, matches
 * from before the window in one round trip, the rest inside LDS, whole rows out.  What the speculation cannot settle from
 * the bytes at hand (length bytes chained beyond 270, a literal run that reaches past the ring, anything wrong) is re-read
 * serially from memory for that one sequence.
 * Round 4's kernel walked the stream with four or five dependent global round trips per sequence and copied a byte per lane
 * (profiles/r4_lz4_counters.txt: 20.7 ms per GiB).  A first round-5 version walked it on the scalar unit out of a register
 * window - ~150 scalar instructions a sequence, and a CU has ONE scalar ALU for its 24 waves: 18.3 ms. */
#define QZK_L4_RING 1024u
#define QZK_L4_Q 128u
#define QZK_L4_SLOW 0xffffffffu
typedef struct __attribute__((aligned(16))) { uint32_t lit_off, lit, ml, off; } qzk_l4rec;
typedef struct {
    const uint8_t *blk;         /* the block's bytes */
    uint32_t n, lim;            /* its size; bytes that may be read from blk (to the end of the frame's input) */
    uint8_t *ring;              /* LDS, QZK_L4_RING + 16: stream byte x at ring[x % RING], the first 16 once more behind the end */
    qzk_l4rec *q;               /* LDS, QZK_L4_Q records */
    uint32_t filled;            /* the ring holds the stream's [filled - RING, filled); a multiple of 512 */
    uint32_t qh, qt;            /* the queue's head and tail (running record numbers) */
    uint32_t s;                 /* where the next sequence begins */
    uint64_t pre;               /* per lane: the 8 bytes at filled + 8 * lane, asked for ahead of time */
} qzk_l4p;
QZ_DEV uint64_t qzk_l4_fetch(const qzk_l4p *P, uint32_t at, int lane)
{
    const uint32_t o = at + 8u * (uint32_t)lane;
    if (o + 8 <= P->lim) return qzk_rb_ld64(P->blk + o);
    uint64_t v = 0;
    for (uint32_t t = 0; t < 8; t++) if (o + t < P->lim) v |= (uint64_t)P->blk[o + t] << (8 * t);
    return v;
}
/* the ring reaches at least 513 bytes beyond s, and s itself is in it */
QZ_DEV void qzk_l4_stage(qzk_l4p *P, int lane)
{
    if (P->s >= P->filled + 512u) { P->filled = P->s & ~511u; P->pre = qzk_l4_fetch(P, P->filled, lane); }    /* a literal run longer than the ring */
    bool put = false;
    while (P->filled <= P->s + 512u) {
        if (!put) qz_lds_sync();                                    /* the lanes have read what this overwrites */
        put = true;
        const uint32_t r = (P->filled & (QZK_L4_RING - 1)) + 8u * (uint32_t)lane;
        *(uint64_t *)(P->ring + r) = P->pre;
        if (r < 16) *(uint64_t *)(P->ring + QZK_L4_RING + r) = P->pre;
        P->filled += 512u;
        P->pre = qzk_l4_fetch(P, P->filled, lane);
    }
    if (put) qz_lds_sync();
}
/* one sequence read serially from memory (wave-uniform): what the window could not settle.  0, or -1: not a sequence */
QZ_DEV int qzk_l4_slow(const uint8_t *blk, uint32_t n, uint32_t *ps, qzk_l4rec *r)
{
    uint32_t p = *ps;
    if (p >= n) return -1;
    const uint32_t tok = blk[p++];
    uint32_t lit = tok >> 4;
    if (lit == 15) { uint32_t b; do { if (p >= n) return -1; b = blk[p++]; lit += b; } while (b == 255); }
    if (lit > n - p) return -1;
    r->lit_off = p; r->lit = lit; r->ml = 0; r->off = 1;
    p += lit;
    if (p == n) { *ps = n; return 0; }                              /* the block's last sequence: literals only */
    if (n - p < 2) return -1;
    const uint32_t off = (uint32_t)blk[p] | (uint32_t)blk[p + 1] << 8; p += 2;
    if (off == 0) return -1;
    uint32_t ml = tok & 15;
    if (ml == 15) { uint32_t b; do { if (p >= n) return -1; b = blk[p++]; ml += b; } while (b == 255); }
    if (p >= n) return -1;                                          /* a block ends with a literals-only sequence */
    r->ml = ml + QZK_LZ4_MINMATCH; r->off = off;
    *ps = p;
    return 0;
}
/* one window: the sequences that begin in [s, s + 64) join the queue.  0, or -1 */
QZ_DEV int qzk_l4_window(qzk_l4p *P, int lane)
{
    qzk_l4_stage(P, lane);
    const uint32_t base = P->s, n = P->n;
    const uint32_t p = base + (uint32_t)lane;
    const uint32_t x = qz_ld32(P->ring + (p & (QZK_L4_RING - 1)));             /* token and the byte behind it */
    const uint32_t tok = x & 0xff;
    uint32_t lit = tok >> 4, ml = tok & 15, q = p + 1;
    bool slow = p >= n;
    if (lit == 15) { const uint32_t b = (x >> 8) & 0xff; lit += b; q++; slow = slow || b == 255; }
    const uint32_t lit_off = q;
    q += lit;
    const bool last = q == n;
    slow = slow || q > n || (!last && q + 3 > P->filled);
    const uint32_t y = qz_ld32(P->ring + (q & (QZK_L4_RING - 1)));             /* offset and the byte behind it (garbage when slow) */
    const uint32_t off = y & 0xffff;
    uint32_t q2 = q + 2;
    if (ml == 15) { const uint32_t b = (y >> 16) & 0xff; ml += b; q2++; slow = slow || (!last && b == 255); }
    ml += QZK_LZ4_MINMATCH;
    slow = slow || (!last && (off == 0 || q2 >= n));                            /* (q2 >= n: a block ends with a literals-only sequence) */
    const uint32_t next = slow ? QZK_L4_SLOW : last ? n : q2;
    /* the chase: from s on, which lanes are sequence starts */
    uint64_t V = 0;
    uint32_t s = base;
    bool hit_slow = false;
    while (s < base + 64u && s < n) {
        const uint32_t i = s - base, nx = qz_readlane(next, (int)i);
        if (nx == QZK_L4_SLOW) { hit_slow = true; break; }
        V |= 1ull << i;
        s = nx;
    }
    if ((V >> lane) & 1) {
        qzk_l4rec r; r.lit_off = lit_off; r.lit = lit; r.ml = last ? 0u : ml; r.off = last ? 1u : off;
        P->q[(P->qt + (uint32_t)qz_popc64(V & qz_below(lane))) & (QZK_L4_Q - 1)] = r;
    }
    P->qt += (uint32_t)qz_popc64(V);
    if (hit_slow) {
        qzk_l4rec r;
        if (qzk_l4_slow(P->blk, n, &s, &r)) return -1;
        if (lane == 0) P->q[P->qt & (QZK_L4_Q - 1)] = r;
        P->qt++;
    }
    P->s = s;
    qz_lds_sync();
    return 0;
}

/* decode one block through S (history = the frame's output so far: linked frames).  0 or an error */
QZ_DEV int qzk_lz4_dblock(qzk_rb *S, const uint8_t *blk, uint32_t n, uint32_t lim, uint8_t *ring, qzk_l4rec *queue, int lane)
{
    if (n == 0) return -1;
    qzk_l4p P; P.blk = blk; P.n = n; P.lim = lim; P.ring = ring; P.q = queue; P.filled = 0; P.qh = P.qt = 0; P.s = 0;
    P.pre = qzk_l4_fetch(&P, 0, lane);
    for (;;) {
        while (P.s < n && P.qt - P.qh < 64u) if (qzk_l4_window(&P, lane)) return -1;
        const uint32_t cnt = P.qt - P.qh < 64u ? P.qt - P.qh : 64u;
        if (cnt == 0) break;                                        /* (the block is through and every sequence resolved) */
        qzk_l4rec r; r.lit_off = 0; r.lit = 0; r.ml = 0; r.off = 1;
        if ((uint32_t)lane < cnt) r = P.q[(P.qh + (uint32_t)lane) & (QZK_L4_Q - 1)];
        const uint32_t span0 = qz_readlane(r.lit_off, 0);
        uint32_t s_tot = qz_wave_incl_scan(r.lit + r.ml);
        const uint32_t s_span = r.lit_off + r.lit - span0;         /* grows with the lane: the literals lie in stream order */
        const uint32_t fit = (uint32_t)qz_popc64(qz_ballot((uint32_t)lane < cnt && s_tot <= QZK_RB_LIM && s_span <= QZK_RB_LITMAX));
        if (fit == 0) {
            /* one sequence that is more than a batch holds: straight to the output */
            const uint32_t gL = qz_readlane(r.lit, 0), gM = qz_readlane(r.ml, 0), gD = qz_readlane(r.off, 0);
            if ((uint64_t)S->obase + gL + gM > S->out_cap) return -1;
            if (gM != 0 && (uint64_t)gD > (uint64_t)S->obase + gL) return -1;
            qzk_rb_flush(S, lane);
            qzk_rb_direct(S, blk + span0, gL, lane);
            qz_wave_sync();
            qzk_rb_skip(S, gL);
            if (gM) { qzk_rb_direct_match(S, gD, gM, lane); qz_wave_sync(); qzk_rb_skip(S, gM); }
            P.qh += 1;
            continue;
        }
        if ((uint32_t)lane >= fit) { s_tot -= r.lit + r.ml; r.lit = 0; r.ml = 0; }
        const uint32_t Tb = qz_readlane(s_tot, (int)fit - 1), span_len = qz_readlane(s_span, (int)fit - 1);
        if (qzk_rb_batch(S, blk + span0, span_len, r.lit_off - span0, r.lit, r.ml, r.off, s_tot, Tb, lane)) return -1;
        P.qh += fit;
    }
    return 0;
}

QZ_KERNEL_OCC(64, 6) qzk_lz4d_kernel(const uint8_t *comp, uint8_t *out, const qzk_lz4seg *segs, qzk_lz4res *res, uint32_t nsegs)
{
    QZ_LDS __attribute__((aligned(16))) uint8_t obuf[QZK_RB_OB];
    QZ_LDS __attribute__((aligned(16))) uint8_t lbuf[QZK_RB_LT];
    QZ_LDS __attribute__((aligned(16))) uint8_t ring[QZK_L4_RING + 16];
    QZ_LDS qzk_l4rec queue[QZK_L4_Q];
    const int lane = qz_lane();
    const uint32_t s = blockIdx.x;
    if (s >= nsegs) return;
    const qzk_lz4seg sg = segs[s];
    const uint8_t *p = comp + sg.in_off;
    uint8_t *o = out + sg.out_off;
    const uint32_t n = sg.in_len;
    int status = QZK_LZ4_EDATA; uint32_t pos = 0;
    qzk_rb S;
    qzk_rb_init(&S, obuf, lbuf, o, 0, sg.out_cap);
    do {
        if (n < 7) { status = QZK_LZ4_EIN; break; }
        if (qz_ld32(p) != 0x184D2204u) break;
        uint32_t flg = p[4];
        if ((flg >> 6) != 1 || (flg & 2)) break;
        const bool bcheck = (flg >> 4) & 1, csize = (flg >> 3) & 1, ccheck = (flg >> 2) & 1, dict = flg & 1;
        /* BD: the frame's largest block (64 KB << 2 (id - 4), id 4..7: 4 MiB at most).  liblz4 refuses a block above it
         * (LZ4F_decompress: maxBlockSizeInvalid / "block size > maxBlockSize"), and so must this decoder: the walk below adds
         * sequence lengths in 32 bits, and only a block of 16 MiB and more could make such a sum wrap */
        const uint32_t bd = p[5], bid = (bd >> 4) & 7;
        if ((bd & 0x8f) || bid < 4) break;
        const uint32_t bmax = 65536u << (2 * (bid - 4));
        pos = 6 + (csize ? 8 : 0) + (dict ? 4 : 0);
        if (pos + 1 > n) { status = QZK_LZ4_EIN; break; }
        if (p[pos] != ((qzk_wave_xxh32(p + 4, pos - 4, lane) >> 8) & 0xff)) break;
        pos++;
        bool ok = true;
        for (;;) {
            if (pos + 4 > n) { status = QZK_LZ4_EIN; ok = false; break; }
            uint32_t bh = qz_ld32(p + pos); pos += 4;
            if (bh == 0) break;
            uint32_t bsz = bh & 0x7fffffffu;
            if (bsz > bmax) { ok = false; break; }
            if (bsz > n - pos) { status = QZK_LZ4_EIN; ok = false; break; }
            if (bh & 0x80000000u) {
                if (bsz > sg.out_cap - S.obase) { status = QZK_LZ4_EOUT; ok = false; break; }
                qzk_rb_flush(&S, lane);
                qzk_rb_direct(&S, p + pos, bsz, lane);
                qz_wave_sync();
                qzk_rb_skip(&S, bsz);
            } else if (qzk_lz4_dblock(&S, p + pos, bsz, n - pos, ring, queue, lane)) { ok = false; break; }
            pos += bsz + (bcheck ? 4 : 0);
        }
        qzk_rb_flush(&S, lane);
        if (!ok) break;
        if (ccheck) {
            if (pos + 4 > n) { status = QZK_LZ4_EIN; break; }
            qz_wave_sync();
            if (qz_ld32(p + pos) != qzk_wave_xxh32_staged(o, S.obase, obuf, lane)) break;
            pos += 4;
        }
        status = QZK_LZ4_OK;
    } while (0);
    if (lane == 0) { qzk_lz4res r; r.status = status; r.in_used = pos; r.out_len = S.obase; r.pad = 0; res[s] = r; }
}

#endif
