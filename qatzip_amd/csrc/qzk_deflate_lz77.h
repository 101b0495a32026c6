/*
 * qzk_deflate_lz77.h — K1: zlib-exact greedy LZ77 parse ("deflate_fast", level 1) of one hw_buff_sz chunk per WAVE
 * (qzk_lz77_chunk), gfx950.  The kernel that runs it - persistent four-wave workgroups, five to a CU, whose waves pull chunks, parse
 * them, code them (K2) and fold their CRC-32 - is qzk_lz77_pull_kernel in qzk_deflate_huff.h.
 *
 * What it replaces: the deflate() hot loop the reference's software path spends
 * ~75 % of its time in (src/qatzip_sw.c:197, zlib deflate_fast + longest_match;
 * restated on the CPU in oracle/qzo_deflate.c).  Output = the exact symbol
 * stream zlib tallies (lit / (len-3, dist)), plus where each 32767-symbol block
 * starts, for K2 (qzk_deflate_huff.h) to Huffman-code.
 *
 * MI355X design:
 *   - the greedy parse is serial by definition; one wave walks the chunk in
 *     WINDOWS of 64 consecutive positions starting at the next parse point.
 *     All 64 lanes speculate in parallel (hash, 4-deep chain walk, 16-byte
 *     candidate compares); a short wave-uniform loop then hops from parse point
 *     to parse point with v_readlane, and only lanes whose hash bucket was also
 *     touched earlier in the same window take a slower exact path.
 *   - zlib's head[] + prev[] chains are kept as one table: per 16-bit hash the four
 *     newest inserted positions (level 1 never follows more than four links) - one
 *     16-byte gather and one scatter per window, no dependent chain reads.  An entry is
 *     4 x 24-bit chunk offsets + a 32-bit epoch (the chunk's number): a new chunk needs
 *     no clear and zlib's window slide no pass over the table.  The four waves of a
 *     workgroup (one chunk each) share table LINES - bucket h of wave w sits next to
 *     bucket h of wave w+1 - so the buckets every chunk of a corpus keeps hitting (its
 *     common trigrams) are a few thousand fully used lines that stay in L2, instead of
 *     four times as many lines with one live entry each.
 *   - LDS (8 KiB per wave) holds a 4 KiB ring of the input - the last ~3.7 KiB and 328 bytes ahead of the window -
 *     from which the lanes' own bytes, three quarters of the candidates and the extension of long matches are
 *     compared, and the per-window slot tables; far candidates are gathered from HBM/L2.  The chunk's CRC-32 is
 *     folded from the same dwords as they enter the ring.
 *   - zlib's window slide (strstart >= 65274 => rebase by 32768, NIL==0) is
 *     reproduced literally, so chunks up to 512 KiB and odd tail sizes match.
 *   - round 6 (profiles/r6_k1_experiments.txt): two things tried against the table traffic stay in this file, compiled out.
 *     QZK_CNBLOG >= 0: a wave keeps the table entries it touched last in LDS (direct-mapped on a multiplicative mix of the
 *     hash, 12 bytes each: the hash + four 16-bit window positions; write-through and always equal to the table, so a hit
 *     answers exactly what the gather would have).  Most positions lie inside matches - text seen before, whose hashes
 *     were looked up a few hundred bytes ago: 256 entries take 44 % of the gathers (model and hardware agree) and 17 % of
 *     the fetched bytes - and nothing of the launch's time.  QZK_PF on top: the next window's entries are asked for while
 *     this window is resolved; the wait for them is 83 clocks a window, and the machinery costs more than the gather's
 *     wait was.  A window is a chain of ~100 dependent LDS / cross-lane / scalar steps; memory is the smaller part of it.
 */
#ifndef QZK_DEFLATE_LZ77_H
#define QZK_DEFLATE_LZ77_H
#include "qzk_common.h"
#include "qzk_crcmath.h"

#define QZK_WSIZE 32768
#define QZK_MAXDIST 32506          /* w_size - MIN_LOOKAHEAD */
#define QZK_MINLOOK 262
#define QZK_LITBUF 32767           /* symbols per block at memLevel 9 */
#define QZK_MAXBLK 20
#define QZK_CAP 16                 /* speculative compare depth (>= nice_match) */
#define QZK_WLIM 61                /* parse points per window: interiors of a len<=4 match stay < 64 */
#define QZK_NICE 8
#define QZK_MAXINS 4
#define QZK_HSIZE 65536            /* zlib hash_bits 16 at memLevel 9 */
#ifndef QZK_CNBLOG
#define QZK_CNBLOG (-1)            /* log2 of the entries of a wave's LDS cache of table entries; -1: none (the default: see the header) */
#endif
#if QZK_CNBLOG >= 0
#define QZK_CNB (1 << QZK_CNBLOG)
#else
#define QZK_CNB 0
#endif
#ifndef QZK_PF
#define QZK_PF (QZK_CNB ? 1 : 0)   /* the next window's table entries are asked for while this window is resolved (needs the cache) */
#endif
#ifndef QZK_PF_MINL
#define QZK_PF_MINL 30             /* ... and a window whose first lane without an entry comes earlier than this asks the table itself */
#endif
#ifndef QZK_NSLOT
#define QZK_NSLOT 256              /* (512 / 256 were round 5's: 3 % fewer lanes on the exact path, and LDS for sixteen waves a CU only) */
#endif
#ifndef QZK_RING
#define QZK_RING 4096              /* bytes of recent input kept in LDS */
#endif
#ifndef QZK_K1_OCC
#define QZK_K1_OCC 5               /* workgroups per CU the register budget is cut for: 5 x 4 waves = five waves a SIMD (96 VGPRs) */
#endif
#define QZK_RINGW (QZK_RING / 4)
#ifndef QZK_NSLOT2                  /* second slot table, keyed by the hash's high bits */
#define QZK_NSLOT2 128              /* (256 without the third table: 1 ms per 4 GiB slower than 128 + 128) */
#endif
#ifndef QZK_NSLOT3                  /* third slot table, keyed by a mix of all the hash's bits (0: none) */
#if QZK_CNB
#define QZK_NSLOT3 0
#else
#define QZK_NSLOT3 128
#endif
#endif
#define QZK_K1_PARSEW (2 * QZK_NSLOT + QZK_RINGW + 4 + QZK_NSLOT2 + QZK_NSLOT3 + 3 * QZK_CNB)   /* words of LDS one wave's parse needs */
#ifndef QZK_K1_WAVES
#define QZK_K1_WAVES 4             /* waves per K1 workgroup, one chunk each, one per SIMD; their entries of a bucket are one 64-byte line
                                    * of the candidate table.  Round 6 (profiles/r6_k1_occupancy.txt): the window is a chain of dependent
                                    * steps, so waves per SIMD are what the rate follows - 4 x 5 workgroups (twenty waves a CU, slot tables
                                    * 256 / 128 / 128 to fit the LDS: a workgroup may take 31 KB - 32 064 B were four workgroups a CU) 105 ms per 4 GiB against 113 for 16 x 1; a workgroup's waves go to the SIMDs
                                    * in turn from SIMD 0, so 10 x 2 leaves a CU with ONE workgroup (3 + 3 waves on a SIMD > 5): 153 ms;
                                    * 4 x 6 (80 VGPRs, 20 spilled) 119 ms */
#endif

/* Where a workgroup's entries lie in the candidate table.  A bucket's four entries of a four-wave workgroup are one 64-byte
 * line (a line another workgroup's waves never read).  The lines of QZK_K1_TABGRP workgroups that run on ONE XCD (workgroups are
 * handed to the eight XCDs in turn, so b, b + 8, b + 16, ... share an L2) lie side by side in a bucket's row: the chunks of a corpus
 * keep hitting the same buckets (its common trigrams), and neighbouring lines share the L2's 128-byte lines and the DRAM's
 * pages.  Measured, one 4 GiB launch (profiles/r6_k1_occupancy.txt): G = 1 (a workgroup's table on its own) 105.2 - 105.8 ms, 2 103.9 - 104.1,
 * 4 104.6, 5 103.6.  Workgroup b: row group (b / (8 G)) * 8 + b % 8, place (b / 8) % G in the row. */
#ifndef QZK_K1_TABGRP
#define QZK_K1_TABGRP 5
#endif
#define QZK_K1_TABW (QZK_K1_WAVES * QZK_K1_TABGRP)
#define QZK_K1_TABROWS(wgs) ((((wgs) + 8u * QZK_K1_TABGRP - 1) / (8u * QZK_K1_TABGRP)) * 8u)      /* row groups (of 65536 rows of QZK_K1_TABW entries) a launch of wgs workgroups needs */

/* one bucket of the candidate table: the four newest inserted positions with this hash, newest first, as 24-bit chunk
 * offsets (0 = none: offset 0 is zlib's NIL), valid only while ep equals the epoch of the chunk being parsed */
typedef struct __attribute__((aligned(16))) { uint32_t w0, w1, w2, ep; } qzk_bkt;
/* moved as ONE 16-byte access (a struct load lets the compiler fetch ep first and the rest behind a branch: two dependent
 * round trips to HBM) */
typedef uint32_t qzk_u32x4 __attribute__((vector_size(16)));

/* The gather of a table entry is served by the L2 (agent-scope `sc1` load), never by this CU's vector L1: the
 * waves of a workgroup keep their entries of one bucket in one cache line, and the L1 (write-through, no write-allocate,
 * not coherent) may be filled with a copy of that line that is already missing a neighbour wave's latest store - which
 * that wave would then read back as its own entry (seen on gfx950 as soon as two waves shared lines).  The L2 is the
 * single home of the line within the XCD, and a wave's loads follow its own stores to it in issue order. */
QZ_DEV qzk_u32x4 qzk_ld_bkt(const qzk_bkt *p)
{
#ifdef QZ_SIM
    return *(const qzk_u32x4 *)p;
#else
    qzk_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
#endif
}

/* The same entry asked for AHEAD of its use (round 6: the next window's entries travel while this window is resolved).
 * The compiler cannot be told: a load it counts is waited for at the next `s_waitcnt vmcnt(0)` it places for any load it
 * cannot count exactly - the candidate compares' and the exact path's conditional ones, every window.  So the load is the
 * kernel's own (the compiler sees the registers as written here) and so is the wait, at the next window's top, before the
 * first read; between the two nothing may touch v: it is only ever passed from the one statement to the other
 * (tools/k1_pf_audit.py checks the built code object for exactly that). */
QZ_DEV void qzk_ld_bkt_ahead(qzk_u32x4 *v, const qzk_bkt *p)
{
#ifdef QZ_SIM
    *v = *(const qzk_u32x4 *)p;
#else
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(*v) : "v"(p) : "memory");
#endif
}
QZ_DEV void qzk_wait_ahead(qzk_u32x4 *v)
{
#ifdef QZ_SIM
    (void)v;
#else
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(*v) : : "memory");
#endif
}

#ifdef QZ_SIM                       /* the emulator's tests want to know that the windows they ran took the new path */
static unsigned long qzk_sim_count[2];         /* [0] windows that asked the table themselves, [1] windows */
#define QZK_SIMC(k) do { if (qz_lane() == 0) qzk_sim_count[k]++; } while (0)
#else
#define QZK_SIMC(k) do { } while (0)
#endif
#if defined(QZK_PROF) && !defined(QZ_SIM)
#define QZK_T(k) do { uint64_t t_ = __builtin_readcyclecounter(); prof[k] += t_ - tprev; tprev = t_; } while (0)
#define QZK_C(k, v) do { prof[k] += (uint64_t)(v); } while (0)
#else
#define QZK_T(k) do { } while (0)
#define QZK_C(k, v) do { } while (0)
#endif

typedef struct {
    uint32_t nsym;                 /* symbols in the chunk */
    uint32_t nfull;                /* blocks closed by the 32767-symbol rule */
    uint32_t bstart[QZK_MAXBLK];   /* chunk-relative byte where block k starts (k <= nfull) */
    uint32_t can_store;            /* bit k: zlib could still emit block k as stored (block_start >= 0) */
    uint32_t n;                    /* chunk length */
#ifdef QZK_PROF
    uint64_t prof[24];             /* per-phase cycles / counters (profiling builds only) */
#endif
} qzk_lzmeta;

/* A call is normally one buffer cut every chunk_sz bytes (only the last chunk is short, only the last can be final).
 * A coalesced launch carries many small requests instead, each starting on a chunk boundary: cdesc[chunk] then gives
 * the chunk's length (bits 0-30) and whether it closes its request's stream (bit 31).  NULL = the normal layout. */
#define QZK_CDESC_FINAL 0x80000000u
QZ_DEV uint32_t qzk_chunk_len(const uint32_t *cdesc, uint32_t chunk, uint64_t src_len, uint32_t chunk_sz)
{
    if (cdesc) return cdesc[chunk] & ~QZK_CDESC_FINAL;
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    return (uint32_t)((src_len - coff) < chunk_sz ? (src_len - coff) : chunk_sz);
}

QZ_DEV uint32_t qzk_ld32g(const uint8_t *src, uint64_t off, uint64_t src_len)
{
    if (off + 4 <= src_len) return qz_ld32(src + off);
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) if (off + k < src_len) v |= (uint32_t)src[off + k] << (8 * k);
    return v;
}

/* the same with one test on the common path */
QZ_DEV uint32_t qzk_ld32g_fast(const uint8_t *src, uint64_t off, uint64_t src_len)
{
    return off + 4 <= src_len ? qz_ld32(src + off) : qzk_ld32g(src, off, src_len);
}

/* low dword of {hi,lo} >> 8*s, s in 0..3 (v_alignbyte_b32) */
QZ_DEV uint32_t qzk_alignbyte(uint32_t hi, uint32_t lo, uint32_t s)
{
#ifdef QZ_SIM
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * s));
#else
    return __builtin_amdgcn_alignbyte(hi, lo, s);
#endif
}

/* common-prefix length of src[a..] and src[b..], at most maxlen; whole wave cooperates */
QZ_DEV int qzk_wave_matchlen(const uint8_t *src, uint64_t src_len, uint64_t a, uint64_t b, int maxlen, int lane)
{
    for (int off = 0; off < maxlen; off += 256) {
        int o = off + 4 * lane;
        bool act = o < maxlen;
        uint32_t x = 0;
        if (act) x = qzk_ld32g(src, a + o, src_len) ^ qzk_ld32g(src, b + o, src_len);
        uint64_t mm = qz_ballot(act && x != 0);
        if (mm) {
            int f = qz_ctz64(mm);
            uint32_t xf = qz_readlane(x, f);
            int len = off + 4 * f + (qz_ctz32(xf) >> 3);
            return len < maxlen ? len : maxlen;
        }
    }
    return maxlen;
}

/* One chunk, one wave.  zlib's head[] + prev[] chains (level 1 never follows more than four links) are kept as ONE
 * table: bkt[hash] = the four most recent inserted positions with that 16-bit hash, newest first - exactly the
 * candidates longest_match() would visit, in its order.  A lookup is a single 16-byte gather instead of a gather plus up
 * to three dependent ones; an insert shifts the entry.  tab points at this wave's column of its workgroup's table (entry
 * h at tab[h * QZK_K1_TABW]); LDS holds, per wave, a ring of the most recent input and the per-window slot tables. */
QZ_DEV void qzk_lz77_chunk(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t chunk,
                           uint8_t *sym_lc, uint16_t *sym_dist, qzk_lzmeta *meta, qzk_bkt *tab, uint32_t epoch,
                           const uint32_t *cdesc, uint32_t *lds, const qzk_k1crc_lds *crcT, uint32_t *crc_slot)
{
    /* this wave's QZK_K1_LDSW words of LDS:
     *   slot[QZK_NSLOT]  per-window: min(lane<<16 | hash) over the lanes on a hash key
     *   scnt[QZK_NSLOT]  per-window: number of lanes on the key
     *   slot2[QZK_NSLOT2] per-window: lowest lane on the hash's HIGH bits - a lane that has an earlier lane with its hash
     *     has one on both keys; half of the lanes the first table alone sent to the exact path had none
     *   slot3[QZK_NSLOT3] the same under a mix of all sixteen bits: with tables of 256 a lane that has NO earlier lane with its hash
     *     still finds an earlier lane on its low byte AND another on its high byte in 0.9 windows out of ten (round 6) - each an entry
     *     into the exact path that leaves at its first test
     *   ring[QZK_RINGW + 4]  the last QZK_RING bytes of input (and ~100 ahead of the parse point).  Every candidate compare
     *     drags a 128-byte line through L2 for 16 bytes, three quarters of them less than 4 KiB back; with a dozen
     *     waves per CU K1 is bound by exactly that traffic (profiles/), so those come from here. */
    uint32_t *const slot = lds, *const scnt = lds + QZK_NSLOT, *const ring = lds + 2 * QZK_NSLOT, *const slot2 = ring + QZK_RINGW + 4;
    uint32_t *const slot3 = slot2 + QZK_NSLOT2; (void)slot3;
#if QZK_CNB
    /* the cache of table entries: ctag[s] = hash | 1 << 16 of the entry held (anything else: none - the commit's election
     * leaves its marks here), cpos[s], cpos[QZK_CNB + s] = its four positions, newest first, as 16-bit offsets from the
     * window origin `base` (0 = none: zlib's NIL; what has slid out of the window is NIL to every later lookup) */
    uint32_t *const ctag = slot3 + QZK_NSLOT3, *const cpos = ctag + QZK_CNB;
#pragma nounroll
    for (uint32_t i = (uint32_t)qz_lane(); i < QZK_CNB; i += 64) ctag[i] = 0;
#endif
    uint32_t rhi = 0;                      /* chunk offset the ring is filled up to (multiple of 256) */
    uint32_t crc_acc = 0;                  /* this lane's share of the chunk's CRC-32 (crcT != NULL) */
    uint32_t rnext = qzk_ld32g_fast(src, (uint64_t)chunk * chunk_sz + 4 * (uint32_t)qz_lane(), src_len);   /* my dword of the row at rhi */
    /* (the ring's first four words are kept a second time behind its last: five words from any place in it are five
     * consecutive words - one address, constant offsets, two words an LDS instruction - instead of five wrapped indices) */
#define QZK_RING16(dst, ca) do { const uint32_t r_ = (ca) & (QZK_RING - 1), i_ = r_ >> 2, s_ = r_ & 3; \
        const uint32_t d0_ = ring[i_], d1_ = ring[i_ + 1], d2_ = ring[i_ + 2], d3_ = ring[i_ + 3], d4_ = ring[i_ + 4]; \
        (dst)[0] = qzk_alignbyte(d1_, d0_, s_); (dst)[1] = qzk_alignbyte(d2_, d1_, s_); \
        (dst)[2] = qzk_alignbyte(d3_, d2_, s_); (dst)[3] = qzk_alignbyte(d4_, d3_, s_); } while (0)
#define QZK_RING4(ca) ({ const uint32_t r_ = (ca) & (QZK_RING - 1), i_ = r_ >> 2; \
        qzk_alignbyte(ring[i_ + 1], ring[i_], r_ & 3); })
#define QZK_AHEAD (64 + 258 + 6)       /* bytes of input the ring holds beyond the window start */

    const int lane = qz_lane();
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const uint32_t n = qzk_chunk_len(cdesc, chunk, src_len, chunk_sz);
    uint8_t *olc = sym_lc;                          /* this chunk's symbol arrays (the caller placed them) */
    uint16_t *odist = sym_dist;
    qzk_lzmeta *mt = meta + chunk;
    /* common-prefix length of the strings at chunk offsets ca (a lane of the current window) and cb < ca, at most maxlen:
     * the whole wave compares, four bytes a lane.  ca's side always comes out of the ring (it reaches QZK_AHEAD past the
     * window), cb's side too unless the candidate is more than ~3.7 KiB back - a match that outgrows the 16 speculative
     * bytes used to cost the parse a round trip to HBM per string */
    auto wave_matchlen = [&](uint32_t ca, uint32_t cb, int maxlen) -> int {
        const bool near = (int64_t)cb >= (int64_t)rhi - QZK_RING;
        for (int off = 0; off < maxlen; off += 256) {
            const int o = off + 4 * lane;
            const bool act = o < maxlen;
            uint32_t x = 0;
            if (act) x = QZK_RING4(ca + (uint32_t)o) ^ (near ? QZK_RING4(cb + (uint32_t)o) : qzk_ld32g(src, coff + cb + (uint32_t)o, src_len));
            const uint64_t mm = qz_ballot(act && x != 0);
            if (mm) {
                const int f = qz_ctz64(mm);
                const uint32_t xf = qz_readlane(x, f);
                const int len = off + 4 * f + (qz_ctz32(xf) >> 3);
                return len < maxlen ? len : maxlen;
            }
        }
        return maxlen;
    };

    /* nothing to clear: the entries of earlier chunks carry other epochs.  (All cross-lane ordering in here is
     * wave-local - the waves of a workgroup never exchange anything - so the syncs are wavefront-scope: no vmcnt(0).) */
    qz_lds_sync();

    uint32_t base = 0;                              /* chunk offset of window position 0 */
    uint32_t fill = n < 65536u ? n : 65536u;        /* chunk offset one past the data zlib has in its window */
    uint32_t avail_in = n - fill;
    uint32_t pos = 0;                               /* next parse point (chunk offset) */
    uint32_t nsym = 0, nfull = 0, cur_bstart = 0, can_store = 0;
    mt->bstart[0] = 0;                              /* wave-uniform values: every lane stores the same word */
#if QZK_PF
    /* entries asked for a window ahead: lane i holds the entry of chunk offset pf_base + i as the table had it BEFORE the
     * commit of the window that asked (pf_okm: lanes that hold one; pf_base = that window's start + its `lim`, where the
     * next one starts unless its last symbol overshoots).  A bucket that commit stores makes the entry held here out of
     * date (only the cache has the new one), and "was my bucket stored" must never be answered with a wrong no: every
     * storing lane leaves pf_mark | bucket in slot[] under the low bits of its hash or, if a lane with another hash got
     * there first, in scnt[] under the high bits (both tables are free between two windows' detection phases); a lane
     * that finds both places taken sets pf_lost, and the next window asks the table itself as every window used to. */
    qzk_u32x4 pf = {0, 0, 0, 0};
    uint64_t pf_okm = 0;
    uint32_t pf_base = 0, pf_mark = 0, wcount = 0;
    bool pf_lost = false;
    /* ... and the window's own stores - its symbols, its table entries - wait in registers until the next window has made
     * its one wait for what was asked ahead: a store issued at the window's end would be the youngest thing that wait
     * waits for (the counter is one for loads and stores), and the store's way to the L2 and back is as long as a load's */
    uint64_t dPm = 0, dSTm = 0;
    uint32_t dnsym = 0, dv_sym = 0, dv_h = 0, dv_e0 = 0, dv_e1 = 0, dv_e2 = 0;
#define QZK_DEFERRED_STORES() do { \
        if ((dPm >> lane) & 1) { const uint32_t i_ = dnsym + (uint32_t)qz_popc64(dPm & qz_below(lane)); \
                                 olc[i_] = (uint8_t)dv_sym; odist[i_] = (uint16_t)(dv_sym >> 8); } \
        if ((dSTm >> lane) & 1) { const qzk_u32x4 e_ = {dv_e0, dv_e1, dv_e2, epoch}; *(qzk_u32x4 *)&tab[(size_t)dv_h * QZK_K1_TABW] = e_; } \
        dPm = 0; dSTm = 0; } while (0)
    bool pf_live = false;
#endif
#if defined(QZK_PROF) && !defined(QZ_SIM)
    uint64_t prof[24] = {0}; uint64_t tprev = __builtin_readcyclecounter();
#endif

    for (;;) {
        /* ---- zlib loop top: fill_window() when lookahead < MIN_LOOKAHEAD ---- */
        uint32_t look = fill - pos;
        if (look < QZK_MINLOOK) {
            if (pos - base >= (uint32_t)(QZK_WSIZE + QZK_MAXDIST)) {
                /* zlib slide_hash(): positions at or below the new origin become NIL - here by the `> base` test of the
                 * lookup (offsets are chunk-absolute), no pass over the table */
                base += QZK_WSIZE;
#if QZK_CNB
                /* the cached positions count from the window origin: move them with it */
                qz_lds_sync();
#pragma nounroll
                for (uint32_t i = (uint32_t)lane; i < 2 * QZK_CNB; i += 64) {
                    /* (signed arithmetic: written as a saturating subtraction of the two halves, this loop crashes the
                     * compiler's instruction selection once it has sixteen trips) */
                    const uint32_t v = cpos[i];
                    const int a = (int)(v & 0xffffu) - QZK_WSIZE, b = (int)(v >> 16) - QZK_WSIZE;
                    cpos[i] = (uint32_t)(a > 0 ? a : 0) | (uint32_t)(b > 0 ? b : 0) << 16;
                }
                qz_lds_sync();
#endif
            }
            if (avail_in) {
                uint32_t more = 65536u - (fill - base);
                uint32_t rd = avail_in < more ? avail_in : more;
                fill += rd; avail_in -= rd;
            }
            look = fill - pos;
            if (look == 0) break;
        }

        QZK_T(0); QZK_C(8, 1);
#if QZK_PF
        qzk_wait_ahead(&pf);                            /* the one wait of a window that need not ask the table itself (unconditional: tools/k1_pf_audit.py follows every path) */
        QZK_T(17);
        QZK_DEFERRED_STORES();
        QZK_T(18);
#endif
        /* ---- speculative phase: all 64 lanes ---- */
        const uint32_t B = pos - base;                  /* window position of lane 0 */
        const uint32_t p = B + (uint32_t)lane;          /* my window position */
        const uint32_t pa = pos + (uint32_t)lane;       /* my chunk offset */
        const int avail = pa < fill ? (int)(fill - pa) : 0;
        const bool canh = avail >= 3;
        /* all of this window's loads stay inside the buffer unless it sits at the very end of it */
        const bool guard = coff + pos + 64 + 2 * QZK_CAP > src_len;
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        {
            /* top the ring up to QZK_AHEAD bytes past the window start (the longest match of the last lane ends before
             * that: extensions beyond the 16 speculative bytes read it from here): one coalesced 256-byte row every few windows.  The row
             * at rhi is always already in a register (rnext, asked for when the row before it was stored): by the time it
             * is needed its load has long landed behind a table gather of an earlier window (loads return in order), so
             * the input stream costs the window no round trip of its own */
            if (rhi < pos + QZK_AHEAD) {
                do {
                    const uint32_t a = rhi + 4 * (uint32_t)lane;
                    const uint32_t ri = (a >> 2) & (QZK_RINGW - 1);
                    ring[ri] = rnext;
                    if (ri < 4) ring[ri + QZK_RINGW] = rnext;
                    /* the chunk's CRC-32 rides along: every byte of the chunk passes here exactly once, lane l seeing the
                     * dwords at 256 r + 4 l; it folds them Horner-style (crc32_combine algebra, 8 LDS lookups a step) */
                    if (crcT && a + 4 <= n) crc_acc = qzk_k1crc_step(crcT, crc_acc, rnext);
                    rhi += 256;
                    rnext = qzk_ld32g_fast(src, coff + a + 256, src_len);
                } while (rhi < pos + QZK_AHEAD);
                qz_lds_sync();
            }
            uint32_t w[4];
            QZK_RING16(w, pa);
            w0 = w[0]; w1 = w[1]; w2 = w[2]; w3 = w[3];
        }
        const uint32_t h = (((w0 & 0xf) << 12) ^ (((w0 >> 8) & 0xff) << 6) ^ ((w0 >> 16) & 0xff)) & 0xffff;
        const uint32_t bucket = h;
        const uint32_t key = h & (QZK_NSLOT - 1);

        /* how far this window's parse may go (wave-uniform) */
        int nvalid = look < 64 ? (int)look : 64;
        int lim = nvalid < QZK_WLIM ? nvalid : QZK_WLIM;
        {   /* stop before a parse point where zlib would slide / refill its window */
            int lstop;
            int first_short = (int)look - (QZK_MINLOOK - 1);       /* first l with lookahead < 262 */
            if (first_short < 0) first_short = 0;
            if (avail_in) lstop = first_short;
            else {
                int sl = (int)(QZK_WSIZE + QZK_MAXDIST) - (int)B;
                lstop = first_short > sl ? first_short : sl;
            }
            if (lstop < 1) lstop = 1;
            if (lim > lstop) lim = lstop;
        }
        QZK_T(1);
        /* candidates as of the window start: one 16-byte gather; first candidate needs dist <= MAX_DIST, chained ones
         * cur_match > limit (zlib's asymmetry), and the chain ends at the first one that fails */
        /* the previous window's table stores are ahead of this gather in the wave's memory stream (same wave, same CU) */
        qz_lds_sync();
        qzk_u32x4 ev = {0, 0, 0, 0};
#if QZK_CNB
        /* the entry may be in the wave's cache: only the lanes that miss ask the table */
        const uint32_t cset = ((h * 40503u) >> (16 - QZK_CNBLOG)) & (QZK_CNB - 1);
        const uint32_t ct = ctag[cset], cp0 = cpos[cset], cp1 = cpos[QZK_CNB + cset];
        const bool chit = canh && ct == (h | 0x10000u);
        QZK_C(8, (uint64_t)qz_popc64(qz_ballot(chit)) << 32);      /* profiling builds: cache hits in the high half of the window count */
#if QZK_PF
        /* what the window before asked for, moved down by the lanes its last symbol overshot; a lane neither the cache nor
         * that serves ends the window three lanes early (the interior of a short match must have its entry too) - unless it
         * comes so early that asking the table now, as every window used to, is the better deal */
        bool evhave = false;
        int limdata = 64;
        {
            const uint32_t sh = pos - pf_base;
            if (pf_live && !pf_lost && sh < 64u - (QZK_PF_MINL + 3)) {
                const int sl = lane + (int)sh;
                const uint32_t a0 = qz_shfl(pf[0], sl), a1 = qz_shfl(pf[1], sl), a2 = qz_shfl(pf[2], sl), a3 = qz_shfl(pf[3], sl);
                if (canh && !chit && sl < 64 && ((pf_okm >> (sl & 63)) & 1) && slot[key] != (pf_mark | h) && scnt[(h >> 8) & (QZK_NSLOT - 1)] != (pf_mark | h)) {
                    ev[0] = a0; ev[1] = a1; ev[2] = a2; ev[3] = a3; evhave = true;
                }
            }
            const bool lack = canh && !chit && !evhave;
            const uint64_t LACK = qz_ballot(lack);
            if (LACK) {
                const int fl = qz_ctz64(LACK);
                if (fl < QZK_PF_MINL + 3) { if (lack) { ev = qzk_ld_bkt(&tab[(size_t)bucket * QZK_K1_TABW]); evhave = true; } QZK_C(15, 1ull << 32); QZK_SIMC(0); }
                else limdata = fl - 3;
            }
        }
#else
        const bool evhave = canh && !chit;
        if (canh && !chit) ev = qzk_ld_bkt(&tab[(size_t)bucket * QZK_K1_TABW]);
#endif
#if QZK_PF
        if (lim > limdata) lim = limdata;
        QZK_SIMC(1);
#endif
        QZK_T(16);
        const bool ev_ok = ev[3] == epoch;                /* another chunk's entry: as good as empty */
        uint32_t q0 = ev_ok ? ev[0] & 0xffffffu : 0, q1 = ev_ok ? (ev[0] >> 24) | ((ev[1] & 0xffffu) << 8) : 0,
                 q2 = ev_ok ? (ev[1] >> 16) | ((ev[2] & 0xffu) << 16) : 0, q3 = ev_ok ? ev[2] >> 8 : 0;   /* chunk offsets, 0 = none */
        if (chit) {
            q0 = (cp0 & 0xffffu) ? base + (cp0 & 0xffffu) : 0; q1 = (cp0 >> 16) ? base + (cp0 >> 16) : 0;
            q2 = (cp1 & 0xffffu) ? base + (cp1 & 0xffffu) : 0; q3 = (cp1 >> 16) ? base + (cp1 >> 16) : 0;
        }
#else
        if (canh) ev = qzk_ld_bkt(&tab[(size_t)bucket * QZK_K1_TABW]);
        const bool ev_ok = ev[3] == epoch;                /* another chunk's entry: as good as empty */
        const uint32_t q0 = ev_ok ? ev[0] & 0xffffffu : 0, q1 = ev_ok ? (ev[0] >> 24) | ((ev[1] & 0xffffu) << 8) : 0,
                       q2 = ev_ok ? (ev[1] >> 16) | ((ev[2] & 0xffu) << 16) : 0, q3 = ev_ok ? ev[2] >> 8 : 0;   /* chunk offsets, 0 = none */
#endif
        /* chained candidates must lie above zlib's `limit` (window coordinates: strstart - MAX_DIST or NIL = 0) */
        const uint32_t lo = base + (p > QZK_MAXDIST ? p - QZK_MAXDIST : 0);
        const int maxlen = avail < 258 ? avail : 258;
        const int nice = avail < QZK_NICE ? avail : QZK_NICE;
        int best_len = 2; uint32_t best_c = 0;
        int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
        bool suspect;
        int nc; uint32_t c0, c1, c2, c3;
        {
            uint32_t x[4][4];
            /* the four candidates' first sixteen bytes.  Round 6's phase clocks (profiles/r6_k1_phases.txt): merely ISSUING these
             * took 2200 clocks a window - four candidates, each behind its own branch on "in the ring or not", each ring
             * read waited for inside its branch.  Now every lane reads the ring for all four places at once, whatever its
             * links are (a dead link points at the lane's own position; an LDS address is always good), one wait; then the
             * lanes whose candidate lies further back than the ring overwrite theirs from memory - predicated on the link
             * being live AND far: the texture path must not carry dead lanes. */
            const bool ok0 = canh && q0 > base && pa - q0 <= QZK_MAXDIST;      /* at or below the window origin: NIL */
            const bool ok1 = ok0 && q1 > lo;
            const bool ok2 = ok1 && q2 > lo;
            const bool ok3 = ok2 && q3 > lo;
            c0 = ok0 ? q0 : pa; c1 = ok1 ? q1 : pa; c2 = ok2 ? q2 : pa; c3 = ok3 ? q3 : pa;
            QZK_RING16(x[0], c0); QZK_RING16(x[1], c1); QZK_RING16(x[2], c2); QZK_RING16(x[3], c3);
#define QZK_LDF(k, ck) do { const uint32_t ca_ = (ck); if ((int64_t)ca_ < (int64_t)rhi - QZK_RING) { const uint64_t g_ = coff + ca_; \
        if (!guard) { x[k][0] = qz_ld32(src + g_); x[k][1] = qz_ld32(src + g_ + 4); x[k][2] = qz_ld32(src + g_ + 8); x[k][3] = qz_ld32(src + g_ + 12); } \
        else { x[k][0] = qzk_ld32g(src, g_, src_len); x[k][1] = qzk_ld32g(src, g_ + 4, src_len); x[k][2] = qzk_ld32g(src, g_ + 8, src_len); x[k][3] = qzk_ld32g(src, g_ + 12, src_len); } } } while (0)
#ifndef QZK_EXP_NOFAR   /* (timing experiment only: wrong bytes) */
            QZK_LDF(0, c0); QZK_LDF(1, c1); QZK_LDF(2, c2); QZK_LDF(3, c3);        /* (a dead link's place is the lane's own: never far) */
#endif
#undef QZK_LDF
            nc = (int)ok0 + (int)ok1 + (int)ok2 + (int)ok3;
            QZK_T(2);
            /* while those loads are in flight: which lanes share a hash with an EARLIER lane of this window?
             * slot[key] <- min(lane<<16 | hash), cnt[key] <- number of lanes on the key.  A lane is clean when it is
             * the lowest lane of its key, or when the key holds exactly two lanes and the other one has a different
             * hash; everything else takes the exact path. */
            const uint32_t key2 = (h >> (16 - 8)) & (QZK_NSLOT2 - 1);
#if QZK_NSLOT3
            const uint32_t key3 = ((h * 40503u) >> 12) & (QZK_NSLOT3 - 1);
            if (canh) slot3[key3] = 0xffffffffu;
#endif
            if (canh) { slot[key] = 0xffffffffu; scnt[key] = 0; slot2[key2] = 0xffffffffu; }
            qz_lds_sync();
            if (canh) { atomicMin(&slot[key], ((uint32_t)lane << 16) | h); atomicAdd(&scnt[key], 1u); atomicMin(&slot2[key2], (uint32_t)lane); }
#if QZK_NSLOT3
            if (canh) atomicMin(&slot3[key3], (uint32_t)lane);
#endif
            qz_lds_sync();
            {
                const uint32_t sv = canh ? slot[key] : 0, sc = canh ? scnt[key] : 0, s2 = canh ? slot2[key2] : 0;
                suspect = canh && (sv >> 16) != (uint32_t)lane && !(sc == 2 && (sv & 0xffff) != h) && s2 != (uint32_t)lane;
#if QZK_NSLOT3
                if (suspect && slot3[key3] == (uint32_t)lane) suspect = false;
#endif
            }
            bool done = false;
            for (int k = 0; k < 4; k++) {
                const uint32_t ck = k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3;
                int len = 0;
                if (k < nc) {
                    uint32_t d;
                    len = QZK_CAP;
                    d = w3 ^ x[k][3]; if (d) len = 12 + (qz_ctz32(d) >> 3);
                    d = w2 ^ x[k][2]; if (d) len = 8 + (qz_ctz32(d) >> 3);
                    d = w1 ^ x[k][1]; if (d) len = 4 + (qz_ctz32(d) >> 3);
                    d = w0 ^ x[k][0]; if (d) len = (qz_ctz32(d) >> 3);
                    if (len > maxlen) len = maxlen;
                    if (!done) {
                        if (len > best_len) { best_len = len; best_c = ck; }
                        if (len >= nice) done = true;
                    }
                }
                if (k == 0) l0 = len; else if (k == 1) l1 = len; else if (k == 2) l2 = len; else l3 = len;
            }
        }
        uint32_t mlen = best_len >= 3 ? (uint32_t)best_len : 0;     /* 0 => literal */
        uint32_t mdist = mlen ? pa - best_c : 0;
        const bool capped = (best_len == QZK_CAP) && (maxlen > QZK_CAP);
        const bool exact0 = nc > 0 && pa - c0 == QZK_MAXDIST;

        QZK_T(3);
        const uint64_t CANH = qz_ballot(canh);
        const uint64_t CAPM = qz_ballot(capped);
        const uint64_t CX = qz_ballot(suspect) | CAPM;
        const uint64_t NOLIT = qz_ballot(mlen != 0) | CX;     /* lanes the literal-run shortcut must stop at */

        QZK_T(4);
        /* ---- serial resolution (wave-uniform) ---- */
#if QZK_PF
        {
            /* ask for the next window's entries now: the resolution below, the symbols and the commit (half of a window's
             * clocks) run while they are on their way.  The parse leaves this window at or behind pos + lim. */
            pf_base = pos + (uint32_t)lim; pf_live = true;
            pf_mark = 0x80000000u | wcount << 16; wcount = (wcount + 1) & 0x7fffu;
            const uint32_t ppa = pf_base + (uint32_t)lane;
            const bool pcanh = ppa + 3 <= fill;
            const uint32_t pw = QZK_RING4(ppa);
            const uint32_t ph = (((pw & 0xf) << 12) ^ (((pw >> 8) & 0xff) << 6) ^ ((pw >> 16) & 0xff)) & 0xffff;
            const uint32_t ps = ((ph * 40503u) >> (16 - QZK_CNBLOG)) & (QZK_CNB - 1);
            const uint32_t pct = ctag[ps], pc0 = cpos[ps], pc1 = cpos[QZK_CNB + ps];
            /* a cached entry in the table's own format (this window's commit may replace it in the cache) */
            const uint32_t g0 = (pc0 & 0xffffu) ? base + (pc0 & 0xffffu) : 0, g1 = (pc0 >> 16) ? base + (pc0 >> 16) : 0,
                           g2 = (pc1 & 0xffffu) ? base + (pc1 & 0xffffu) : 0, g3 = (pc1 >> 16) ? base + (pc1 >> 16) : 0;
            pf[0] = g0 | (g1 << 24); pf[1] = (g1 >> 8) | (g2 << 16); pf[2] = (g2 >> 16) | (g3 << 8); pf[3] = epoch;
            if (pcanh && pct != (ph | 0x10000u)) qzk_ld_bkt_ahead(&pf, &tab[(size_t)ph * QZK_K1_TABW]);
            pf_okm = qz_ballot(pcanh);
        }
#endif
        uint64_t Pm = 0;
        uint64_t SHm = qz_ballot(mlen >= 3 && mlen <= QZK_MAXINS && avail - (int)mlen >= 3), S4m = qz_ballot(mlen == 4);
        int l = 0;
        while (l < lim) {
            uint64_t cxr = CX >> l;
            int nextc = cxr ? l + qz_ctz64(cxr) : 64;
            int stop = nextc < lim ? nextc : lim;
            while (l < stop) {                      /* hot loop: hop over clean parse points */
                /* one trip = a run of literal lanes (each of them a parse point, no readlane needed) plus the match
                 * lane that ends it.  The scalar unit is what sixteen waves per CU fight over, so this loop is kept
                 * to a dozen scalar instructions per trip: bit 63 of the mask is a sentinel (no zero test), the
                 * readlane is unconditional (its index is always a lane of the window), lim <= 61 keeps the
                 * mask arithmetic clear of 64-bit shifts by 64. */
                const uint64_t nl = (NOLIT | (1ull << 63)) >> l;
                int e = l + qz_ctz64(nl);           /* first lane at or after l that is not a plain literal */
                if (e > stop) e = stop;
                const uint32_t ml = qz_readlane(mlen, e);
                const int isM = e < stop ? 1 : 0;   /* inside [l, stop) that lane is a clean match (complex lanes end the range) */
                Pm |= ((1ull << (e - l + isM)) - 1) << l;
                l = e + (isM ? (int)ml : 0);
            }
            if (l >= lim || l != nextc) continue;
            /* -- exact path for lane l -- */
            QZK_T(5); QZK_C(9, 1);
            {
                const uint32_t h_l = qz_readlane(h, l);
                /* early out: no earlier lane of this window carries the hash and nothing needs extending =>
                 * the speculative answer is already exact */
                const uint64_t earlier = qz_ballot(canh && h == h_l) & qz_below(l);
                const uint32_t ml_l = qz_readlane(mlen, l);
                if (earlier == 0 && !((CAPM >> l) & 1)) {
                    QZK_C(15, 1);
                    Pm |= 1ull << l;
                    l += ml_l ? (int)ml_l : 1;
                    QZK_T(6);
                    continue;
                }
                const int avail_l = (int)look - l;
                const int maxlen_l = avail_l < 258 ? avail_l : 258;
                const int nice_l = avail_l < QZK_NICE ? avail_l : QZK_NICE;
                int cnt = 0, bl = 2; uint32_t bd = 0; bool fin = false;
                bool had_intra = false;
                if (earlier) {
                    /* inserted lanes so far: parse points with >=3 bytes ahead + interiors of short matches (SHm / S4m: the
                     * lanes whose match is short / exactly four long, kept up to date as exact lanes change theirs) */
                    const uint64_t ps = Pm & SHm;
                    const uint64_t I = (Pm & CANH) | (ps << 1) | (ps << 2) | ((ps & S4m) << 3);
                    uint64_t Sh = earlier & I;
                    if (B == 0) Sh &= ~1ull;               /* window position 0 is NIL */
                    had_intra = Sh != 0;
                    if (Sh) {
                        const uint32_t a0 = qz_readlane(w0, l), a1 = qz_readlane(w1, l), a2 = qz_readlane(w2, l), a3 = qz_readlane(w3, l);
                        while (Sh && cnt < 4 && !fin) {
                            int j = qz_msb64(Sh);
                            Sh &= ~(1ull << j);
                            /* both strings start inside this window: their first 16 bytes are already in registers */
                            int len = QZK_CAP;
                            {
                                uint32_t d;
                                d = a3 ^ qz_readlane(w3, j); if (d) len = 12 + (qz_ctz32(d) >> 3);
                                d = a2 ^ qz_readlane(w2, j); if (d) len = 8 + (qz_ctz32(d) >> 3);
                                d = a1 ^ qz_readlane(w1, j); if (d) len = 4 + (qz_ctz32(d) >> 3);
                                d = a0 ^ qz_readlane(w0, j); if (d) len = (qz_ctz32(d) >> 3);
                            }
                            if (len > maxlen_l) len = maxlen_l;
                            if (len == QZK_CAP && maxlen_l > QZK_CAP)
                                len = wave_matchlen(pos + (uint32_t)l, pos + (uint32_t)j, maxlen_l);
                            cnt++;
                            if (len > bl) { bl = len; bd = (uint32_t)(l - j); }
                            if (len >= nice_l) fin = true;
                        }
                    }
                }
                const int nc_l = (int)qz_readlane((uint32_t)nc, l);
                const bool ex_l = qz_readlane((uint32_t)exact0, l) != 0;
                if (!fin && !(had_intra && ex_l)) {
                    for (int k = 0; k < nc_l && cnt < 4 && !fin; k++) {
                        const uint32_t ck = qz_readlane(k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3, l);
                        int len = (int)qz_readlane((uint32_t)(k == 0 ? l0 : k == 1 ? l1 : k == 2 ? l2 : l3), l);
                        if (len == QZK_CAP && maxlen_l > QZK_CAP)
                            len = wave_matchlen(pos + (uint32_t)l, ck, maxlen_l);
                        cnt++;
                        if (len > bl) { bl = len; bd = (pos + (uint32_t)l) - ck; }
                        if (len >= nice_l) fin = true;
                    }
                }
                uint32_t nl = bl >= 3 ? (uint32_t)bl : 0;
                if (lane == l) { mlen = nl; mdist = nl ? bd : 0; }
                {
                    const uint64_t bit = 1ull << l;
                    SHm = (nl >= 3 && nl <= QZK_MAXINS && avail_l - (int)nl >= 3) ? SHm | bit : SHm & ~bit;
                    S4m = nl == 4 ? S4m | bit : S4m & ~bit;
                }
                Pm |= 1ull << l;
                l += nl ? (int)nl : 1;
            }
            QZK_T(6);
        }

        QZK_T(5);
        /* ---- vector epilogue: symbols, block marks, table commit ---- */
        const uint64_t ps = Pm & SHm;
        const uint64_t I = (Pm & CANH) | (ps << 1) | (ps << 2) | ((ps & S4m) << 3);
        const bool isP = (Pm >> lane) & 1, isI = (I >> lane) & 1;

        const uint32_t rank = (uint32_t)qz_popc64(Pm & qz_below(lane));
        const uint32_t idx = nsym + rank;
        const uint32_t step = mlen ? mlen : 1;
        if (isP) {
#if QZK_PF
            dv_sym = (mlen ? mlen - 3 : (w0 & 0xff)) | mdist << 8;
#else
            olc[idx] = (uint8_t)(mlen ? mlen - 3 : (w0 & 0xff));
            odist[idx] = (uint16_t)mdist;
#endif
        }
        {   /* a symbol that completes a 32767-symbol block (at most one per window) */
            const bool closes = isP && ((idx + 1) % QZK_LITBUF == 0);
            const uint64_t cm = qz_ballot(closes);
            if (cm) {
                int f = qz_ctz64(cm);
                uint32_t nb = qz_readlane(pa + step, f);
                if (cur_bstart >= base) can_store |= 1u << nfull;
                nfull++;
                cur_bstart = nb;
                if (nfull < QZK_MAXBLK) mt->bstart[nfull] = nb;
            }
        }
#if QZK_PF
        dPm = Pm; dnsym = nsym;
#endif
        nsym += (uint32_t)qz_popc64(Pm);

        QZK_T(7);
        /* table commit, ONE scatter per window (the texture path is what K1 saturates: every extra store instruction
         * counts).  A clean inserted lane puts its position in front of the entry seen at window start (the oldest one
         * drops out).  Suspect inserted lanes, in position order: the entry is [me, the (up to three) most recent
         * inserted lanes of this window with my hash, then what the table held at window start], and every earlier
         * lane with that hash is superseded - only the last writer of a hash stores. */
        uint32_t n0 = pa, n1 = q0, n2 = q1, n3 = q2;       /* clean lane: me, then what the table held at window start */
        bool store = isI;
        {
            uint64_t todo = I & qz_ballot(suspect);
            QZK_C(10, qz_popc64(todo)); QZK_C(11, qz_popc64(Pm));
            while (todo) {
                const int j = qz_msb64(todo);               /* the last inserted lane of its bucket: it writes the entry ... */
                uint32_t b_j = qz_readlane(bucket, j);
                const uint64_t same = qz_ballot(canh && bucket == b_j) & I;
                todo &= ~same;                              /* ... and stands for every earlier one */
                uint64_t mates = same & qz_below(j);
                uint32_t m1 = 0, m2 = 0, m3 = 0;            /* up to three most recent inserted lanes with lane j's hash */
                int nf = 1;
                while (mates && nf < 4) {
                    const int m = qz_msb64(mates);
                    mates &= ~(1ull << m);
                    const uint32_t mp = pos + (uint32_t)m;
                    if (nf == 1) m1 = mp; else if (nf == 2) m2 = mp; else m3 = mp;
                    nf++;
                }
                if (lane == j) {
                    if (nf == 2) { n1 = m1; n2 = q0; n3 = q1; }
                    else if (nf == 3) { n1 = m1; n2 = m2; n3 = q0; }
                    else if (nf == 4) { n1 = m1; n2 = m2; n3 = m3; }
                }
                if (((same >> lane) & 1) && lane < j) store = false;
            }
        }
#if QZK_PF
        dSTm = qz_ballot(store); dv_h = bucket;
        dv_e0 = n0 | (n1 << 24); dv_e1 = (n1 >> 8) | (n2 << 16); dv_e2 = (n2 >> 16) | (n3 << 8);
#else
        if (store) {
            const qzk_u32x4 e = {n0 | (n1 << 24), (n1 >> 8) | (n2 << 16), (n2 >> 16) | (n3 << 8), epoch};
            *(qzk_u32x4 *)&tab[(size_t)bucket * QZK_K1_TABW] = e;
        }
#endif
#if QZK_CNB
        {
            /* the cache follows the table: a lane that stores its bucket's new entry puts it into the cache as well, and a
             * lane that had to ask the table leaves the answer there (its bucket unchanged by this window - had any lane
             * of the window entered the hash, that bucket's storing lane would be on the same cache entry and go first).
             * One winner per cache entry: the lanes mark it - the askers, then the storing lanes over them - and who
             * reads his own mark back writes all three words.  An entry that is marked is always rewritten, so whatever
             * the cache holds afterwards is what the table holds. */
            const bool ralloc = evhave && !isI;
            const uint32_t mark = (store ? 0x20040u : 0x20000u) | (uint32_t)lane;
#if QZK_PF
            {   /* what was asked ahead for a stored bucket is out of date now */
                const uint32_t mk = pf_mark | h, key3 = (h >> 8) & (QZK_NSLOT - 1);
                if (store) slot[key] = mk;
                qz_lds_sync();
                const bool second = store && slot[key] != mk;
                if (second) scnt[key3] = mk;
                qz_lds_sync();
                pf_lost = qz_ballot(second && scnt[key3] != mk) != 0;
            }
#endif
            if (ralloc) ctag[cset] = mark;
            qz_lds_sync();
            if (store) ctag[cset] = mark;
            qz_lds_sync();
            if ((store || ralloc) && ctag[cset] == mark) {
                const uint32_t e0 = store ? n0 : q0, e1 = store ? n1 : q1, e2 = store ? n2 : q2, e3 = store ? n3 : q3;
                const uint32_t r0 = e0 > base ? e0 - base : 0, r1 = e1 > base ? e1 - base : 0, r2 = e2 > base ? e2 - base : 0,
                               r3 = e3 > base ? e3 - base : 0;
                cpos[cset] = r0 | r1 << 16; cpos[QZK_CNB + cset] = r2 | r3 << 16;
                ctag[cset] = h | 0x10000u;
            }
        }
#endif
        pos += (uint32_t)l;
        QZK_T(12);
    }

#if QZK_PF
    qzk_wait_ahead(&pf);                                /* nothing of this chunk's may still be on its way into a register */
    QZK_DEFERRED_STORES();
#undef QZK_DEFERRED_STORES
#endif
    if (crcT) {
        /* a last match may have carried the parse to the end of the chunk past rows the ring never asked for */
        for (; rhi < n; rhi += 256) {
            const uint32_t a = rhi + 4 * (uint32_t)lane;
            if (a + 4 <= n) crc_acc = qzk_k1crc_step(crcT, crc_acc, rnext);
            rnext = qzk_ld32g_fast(src, coff + a + 256, src_len);
        }
        /* a lane's dwords end at 4 l + 4 + 256 (rows - 1); shift its share over the dword-covered bytes that follow, XOR the shares, and
         * append the n & 3 bytes no dword covered (crc32_combine with the CRC of that tail) */
        const uint32_t n4 = n & ~3u;
        const uint32_t mine = n4 > 4u * (uint32_t)lane ? (n4 - 4u * (uint32_t)lane + 252u) >> 8 : 0;   /* dwords this lane folded */
        uint32_t part = 0;
        if (mine) part = qzk_multmodp(qzk_x2nmodp(crcT->x2n, n4 - (4u * (uint32_t)lane + 4u + 256u * (mine - 1)), 3), crc_acc);
        for (int d = 32; d >= 1; d >>= 1) part ^= qz_shfl(part, lane ^ d);
        if (n & 3u) {
            uint32_t c = 0xffffffffu;
            for (uint32_t i = n4; i < n; i++) c = qzk_crc_byte(c, src[coff + i]);
            part = qzk_multmodp(qzk_x2nmodp(crcT->x2n, n & 3u, 3), part) ^ ~c;
        }
        *crc_slot = part;                   /* wave-uniform: every lane stores the same word */
    }
    /* zlib's final loop top (lookahead == 0) may still slide before the last flush */
    if (cur_bstart >= base) can_store |= 1u << nfull;
    mt->nsym = nsym; mt->nfull = nfull; mt->can_store = can_store; mt->n = n;      /* uniform, all lanes */
#if defined(QZK_PROF) && !defined(QZ_SIM)
    for (int k = 0; k < 24; k++) mt->prof[k] = prof[k];
#endif
#undef QZK_RING16
#undef QZK_RING4
}

#endif
