/*
 * qzk_lz_batch.h — the LZ77 copy engine shared by the inflate's phase B (qzk_lz_resolve_kernel, qzk_inflate_lane.h) and
 * the LZ4 frame decoder (qzk_lz4d2_kernel, qzk_lz4.h): ONE WAVE resolves up to 64 sequences {literal run, match length,
 * distance} at a time into its segment's output.  It stands where zlib's inflate_fast() window copy
 * (src/qatzip_sw.c:339) and LZ4F_decompress()'s (src/qatzip_sw.c:506) stand in the reference's software path.
 *
 * Round 5 rewrite.  What the counters of the round-4 kernel said (profiles/r4_decode_counters.txt): 17 vector-memory
 * instructions and ~1.0 VALU + 0.7 SALU instructions per output byte; half of the stores were single bytes at the two ragged
 * ends of every batch, the literals arrived a byte per lane behind a six-step cross-lane binary search, 23 VGPRs lived in
 * scratch.  Here a batch is put together in LDS and only WHOLE 16-byte rows ever leave it:
 *   - the wave keeps a window `ob` whose byte 0 is a 16-byte-aligned address of the output; a batch's ragged last row
 *     stays in the window (moved to row 0) and leaves with the next batch - a row of the output is stored once, 16 bytes
 *     wide, by one lane; only a segment's first and last row (shared with the neighbouring segments' waves) are stored
 *     under a byte mask, sixteen lanes one instruction;
 *   - the batch's literals (contiguous in the literal stream / inside one span of an LZ4 block) are fetched as aligned
 *     16-byte rows, one instruction, into an LDS staging area, and every sequence's lane deals its own run out LDS to LDS
 *     in 8-byte steps;
 *   - a match is split where its source crosses the window's start: the part that comes from before the window is final
 *     in memory (the rows were stored earlier by this very wave - its loads follow its stores to the L2 in issue order) and
 *     all lanes' parts are asked for together, at most four 8-byte loads a lane, in flight beside the literal rows: ONE
 *     exposed round trip per batch; the part that comes from the window itself is copied inside LDS in dependency order,
 *     short periods as a replicated 8-byte pattern;
 *   - the next batch's sequence records are asked for before this batch is worked on (by the callers).
 * Wave-uniform control flow throughout (the CPU emulator of tests/sim checks exactly that).
 */
#ifndef QZK_LZ_BATCH_H
#define QZK_LZ_BATCH_H
#include "qzk_common.h"

#ifndef QZK_RB_LIM
#define QZK_RB_LIM 3072            /* output bytes one batch may produce (a batch of 64 sequences is ~350 on the bench data) */
#endif
#define QZK_RB_LITMAX 992          /* bytes of literal span one batch may stage: 64 rows of 16 with any phase */
#define QZK_RB_OB (QZK_RB_LIM + 48)
#define QZK_RB_LT (1024 + 16)
#ifndef QZK_RB_COOP
#define QZK_RB_COOP 32             /* copies inside the window at least this long are made by the whole wave */
#endif
#define QZK_RB_LITCOOP 64

#ifdef QZ_SIM
#define QZK_NOUNROLL
#else
#define QZK_NOUNROLL _Pragma("clang loop unroll(disable)")
#endif
typedef uint32_t qzk_rb_u32x4 __attribute__((vector_size(16)));
typedef struct __attribute__((packed, aligned(1))) { uint64_t v; } qzk_rb_u64u;
QZ_DEV uint64_t qzk_rb_ld64(const uint8_t *p) { return ((const qzk_rb_u64u *)p)->v; }
QZ_DEV void qzk_rb_st64(uint8_t *p, uint64_t v) { ((qzk_rb_u64u *)p)->v = v; }
QZ_DEV void qzk_rb_st32(uint8_t *p, uint32_t v) { ((qz_u32u *)p)->v = v; }
QZ_DEV void qzk_rb_st16(uint8_t *p, uint32_t v) { ((qz_u16u *)p)->v = (uint16_t)v; }

typedef struct {
    uint8_t *ob, *lt;              /* LDS: the output window (QZK_RB_OB bytes, 16-byte aligned), the literal staging (QZK_RB_LT) */
    uint8_t *o;                    /* the segment's output */
    uint64_t hist;                 /* bytes a match may reach back before o */
    uint32_t out_cap;
    uint32_t oph;                  /* o's 16-byte phase */
    uint32_t obase;                /* output bytes of the batches so far */
    uint32_t hd;                   /* leading bytes of the window's row 0 that are NOT this wave's to store: the segment's
                                    * first row until it has left, and whatever went to memory directly (flushes, stored blocks) */
} qzk_rb;

QZ_DEV void qzk_rb_init(qzk_rb *S, uint8_t *ob, uint8_t *lt, uint8_t *o, uint64_t hist, uint32_t out_cap)
{
    S->ob = ob; S->lt = lt; S->o = o; S->hist = hist; S->out_cap = out_cap;
    S->oph = (uint32_t)((uintptr_t)o & 15); S->obase = 0; S->hd = S->oph;
}
/* the window's byte 0 is output position obase - sh; what the window holds begins at wstart */
QZ_DEV uint32_t qzk_rb_sh(const qzk_rb *S) { return (S->oph + S->obase) & 15; }

/* n bytes, LDS to LDS, no overlap */
QZ_DEV void qzk_rb_copy(uint8_t *d, const uint8_t *s, uint32_t n)
{
    if (n >= 8) {
        uint32_t i = 0;
        QZK_NOUNROLL
        for (; i + 8 <= n; i += 8) qzk_rb_st64(d + i, qzk_rb_ld64(s + i));
        if (i < n) qzk_rb_st64(d + n - 8, qzk_rb_ld64(s + n - 8));
    } else {
        if (n & 4) { qzk_rb_st32(d, qz_ld32(s)); d += 4; s += 4; }
        if (n & 2) { qzk_rb_st16(d, qz_ld16(s)); d += 2; s += 2; }
        if (n & 1) *d = *s;
    }
}
/* the classic overlapping copy inside LDS: n bytes to d from d - dist.  A wave's DS operations execute in order, so a load
 * sees this lane's earlier stores. */
QZ_DEV void qzk_rb_copy_back(uint8_t *d, uint32_t dist, uint32_t n)
{
    const uint8_t *s = d - dist;
    if (dist >= 8) { qzk_rb_copy(d, s, n); return; }        /* 8-byte steps never read what they have not written yet */
    /* period below 8: the period, repeated to 8 bytes in a register; stored at multiples of the period it is right wherever
     * it lands, so the steps are the largest multiple of the period that fits 8 bytes */
    uint64_t q = qzk_rb_ld64(s) & ((1ull << (8 * dist)) - 1);       /* (the bytes behind the period are masked off; the window has room to read them) */
    q |= q << (8 * dist);
    if (2 * dist < 8) { q |= q << (16 * dist); if (4 * dist < 8) q |= q << (32 * dist); }
    const uint32_t step = dist == 3 ? 6 : dist == 5 ? 5 : dist == 6 ? 6 : dist == 7 ? 7 : 8;
    uint32_t i = 0;
    QZK_NOUNROLL
    for (; i + 8 <= n; i += step) qzk_rb_st64(d + i, q);
    /* i is a multiple of the period: the rest are the pattern's first bytes */
    uint32_t r = n - i;
    uint8_t *t = d + i;
    if (r >= 8) { qzk_rb_st64(t, q); return; }               /* (not reached: the loop leaves fewer than 8) */
    if (r & 4) { qzk_rb_st32(t, (uint32_t)q); t += 4; q >>= 32; }
    if (r & 2) { qzk_rb_st16(t, (uint32_t)q); t += 2; q >>= 16; }
    if (r & 1) *t = (uint8_t)q;
}

/* the window's pending bytes (its ragged row 0) go to memory under a byte mask: before anything that writes the output
 * directly, and at the segment's end */
QZ_DEV void qzk_rb_flush(qzk_rb *S, int lane)
{
    const uint32_t sh = qzk_rb_sh(S);
    qz_lds_sync();
    if ((uint32_t)lane >= S->hd && (uint32_t)lane < sh) (S->o + S->obase - sh)[lane] = S->ob[lane];
    S->hd = sh;
}
/* after `n` bytes went to the output directly (behind a flush): the window is empty and begins inside a row */
QZ_DEV void qzk_rb_skip(qzk_rb *S, uint32_t n) { S->obase += n; S->hd = qzk_rb_sh(S); }

/* How many leading sequences of the wave's 64 fit one batch: inclusive sums of output bytes and of literal-span bytes. */
QZ_DEV uint32_t qzk_rb_fit(uint32_t s_tot, uint32_t s_span)
{
    return (uint32_t)qz_popc64(qz_ballot(s_tot <= QZK_RB_LIM && s_span <= QZK_RB_LITMAX));
}

/* One batch.  Per lane: a sequence (litrun literals, then mlen bytes from dist back; zeros for idle lanes); lsrc = where the
 * lane's literals begin inside the span [span, span + span_len) of device memory that holds all the batch's literals.
 * s_tot = the inclusive scan of litrun + mlen over the lanes (the caller has it from qzk_rb_fit).  Preconditions: the wave's
 * total <= QZK_RB_LIM, span_len <= QZK_RB_LITMAX.  Returns 0, QZK_INF_EOUT (-2) or QZK_INF_EHIST (-4); wave-uniform. */
QZ_DEV int qzk_rb_batch(qzk_rb *S, const uint8_t *span, uint32_t span_len, uint32_t lsrc, uint32_t litrun, uint32_t mlen,
                        uint32_t dist, uint32_t s_tot, uint32_t Tb, int lane)
{
    uint8_t *const ob = S->ob;
    uint8_t *const lt = S->lt;
    const uint32_t obase = S->obase, sh = qzk_rb_sh(S);
    const uint32_t wstart = obase - sh + S->hd;                    /* first output position the window holds */
    const uint32_t my_o = obase + s_tot - (litrun + mlen), my_m = my_o + litrun;
    if ((uint64_t)obase + Tb > S->out_cap) return -2;
    if (qz_ballot(mlen != 0 && (uint64_t)dist > (uint64_t)my_m + S->hist) != 0) return -4;
#define QZK_OBI(p) ((uint32_t)(p) - obase + sh)
    /* ---- loads: the literal rows and every match's part from before the window, all in flight together ---- */
    const uint32_t ph = (uint32_t)((uintptr_t)span & 15);
    const uint32_t rows = span_len ? (ph + span_len + 15) >> 4 : 0;
    qzk_rb_u32x4 lrow = {0, 0, 0, 0};
    if ((uint32_t)lane < rows) lrow = *(const qzk_rb_u32x4 *)(span - ph + 16 * (uint32_t)lane);
    const uint32_t gap = my_m - wstart;                            /* bytes of the window below my match */
    const uint32_t nmem = (mlen != 0 && dist > gap) ? (mlen < dist - gap ? mlen : dist - gap) : 0;
    const uint8_t *const ms = S->o + ((int64_t)my_m - (int64_t)dist);
    uint64_t a = 0, b = 0, c = 0, e = 0;
    if (nmem != 0 && nmem <= 32) {
        if (nmem >= 8) {
            a = qzk_rb_ld64(ms); b = qzk_rb_ld64(ms + nmem - 8);
            if (nmem > 16) { c = qzk_rb_ld64(ms + 8); if (nmem > 24) e = qzk_rb_ld64(ms + 16); }
        } else if (nmem >= 4) { a = qz_ld32(ms); b = qz_ld32(ms + nmem - 4); }
        else { a = nmem >= 2 ? qz_ld16(ms) : (uint32_t)ms[0]; if (nmem == 3) b = ms[2]; }
    }
    qz_lds_sync();                                                  /* the rows of the batch before have been read out */
    if ((uint32_t)lane < rows) *(qzk_rb_u32x4 *)(lt + 16 * (uint32_t)lane) = lrow;
    if (nmem != 0 && nmem <= 32) {
        uint8_t *d = ob + QZK_OBI(my_m);
        if (nmem >= 8) {
            qzk_rb_st64(d, a);
            if (nmem > 16) { qzk_rb_st64(d + 8, c); if (nmem > 24) qzk_rb_st64(d + 16, e); }
            qzk_rb_st64(d + nmem - 8, b);
        } else if (nmem >= 4) { qzk_rb_st32(d, (uint32_t)a); qzk_rb_st32(d + nmem - 4, (uint32_t)b); }
        else { if (nmem >= 2) qzk_rb_st16(d, (uint32_t)a); else *d = (uint8_t)a; if (nmem == 3) d[2] = (uint8_t)b; }
    }
    {   /* long parts from before the window: the whole wave, 8 bytes a lane */
        uint64_t wide = qz_ballot(nmem > 32);
        while (wide) {
            const int g = qz_ctz64(wide);
            wide &= wide - 1;
            const uint32_t M = qz_readlane(my_m, g), D = qz_readlane(dist, g), N = qz_readlane(nmem, g);
            const uint8_t *s = S->o + ((int64_t)M - (int64_t)D);
            uint8_t *d = ob + QZK_OBI(M);
            for (uint32_t i = 8 * (uint32_t)lane; i + 8 <= N; i += 512) qzk_rb_st64(d + i, qzk_rb_ld64(s + i));
            if (lane == 0 && (N & 7)) qzk_rb_st64(d + N - 8, qzk_rb_ld64(s + N - 8));
        }
    }
    qz_lds_sync();
    /* ---- literals: LDS to LDS, every sequence's lane its own run ---- */
    if (litrun != 0 && litrun < QZK_RB_LITCOOP) qzk_rb_copy(ob + QZK_OBI(my_o), lt + ph + lsrc, litrun);
    {
        uint64_t wide = qz_ballot(litrun >= QZK_RB_LITCOOP);
        while (wide) {
            const int g = qz_ctz64(wide);
            wide &= wide - 1;
            const uint32_t O = qz_readlane(my_o, g), L = qz_readlane(litrun, g), X = qz_readlane(lsrc, g);
            const uint8_t *s = lt + ph + X;
            uint8_t *d = ob + QZK_OBI(O);
            for (uint32_t i = 8 * (uint32_t)lane; i + 8 <= L; i += 512) qzk_rb_st64(d + i, qzk_rb_ld64(s + i));
            if (lane == 0 && (L & 7)) qzk_rb_st64(d + L - 8, qzk_rb_ld64(s + L - 8));
        }
    }
    qz_lds_sync();
    /* ---- what matches read from the window itself: dependency order ---- */
    {
        const uint32_t rem = mlen - nmem, mst = my_m + nmem;        /* bytes left to copy, where they land */
        const uint32_t src_end = mst - dist + (rem < dist ? rem : dist);
        uint64_t pending = qz_ballot(rem != 0);
        while (pending) {
            /* everything below the first unfinished copy is final: it and every copy reading only from there go now */
            const int f = qz_ctz64(pending);
            const uint32_t m_f = qz_readlane(mst, f);
            const bool ready = ((pending >> lane) & 1) && (lane == f || src_end <= m_f);
            uint64_t wide = qz_ballot(ready && rem >= QZK_RB_COOP);
            if (ready && rem < QZK_RB_COOP) qzk_rb_copy_back(ob + QZK_OBI(mst), dist, rem);
            while (wide) {
                const int g = qz_ctz64(wide);
                wide &= wide - 1;
                const uint32_t M = qz_readlane(mst, g), D = qz_readlane(dist, g), L = qz_readlane(rem, g);
                /* the source period [M - D, M) is final: byte i of the copy is byte i mod D of it */
                const uint32_t stp = 64u % D;
                uint32_t r = (uint32_t)lane % D;
                const uint8_t *s = ob + QZK_OBI(M - D);
                uint8_t *d = ob + QZK_OBI(M);
                for (uint32_t i = (uint32_t)lane; i < L; i += 64) {
                    d[i] = s[r];
                    r += stp; if (r >= D) r -= D;
                }
            }
            pending &= ~qz_ballot(ready);
            qz_lds_sync();
        }
    }
    /* ---- out: whole rows; the ragged last one moves to row 0 and waits for the next batch ---- */
    {
        const uint32_t total = sh + Tb, R = total >> 4;
        if (R) {
            uint8_t *const g0 = S->o + obase - sh;                  /* 16-byte aligned */
            uint32_t first = 0;
            if (S->hd) {                                            /* the segment's first row: the bytes before it are another wave's */
                if ((uint32_t)lane >= S->hd && lane < 16) g0[lane] = ob[lane];
                first = 1; S->hd = 0;
            }
            for (uint32_t rw = first + (uint32_t)lane; rw < R; rw += 64)
                *(qzk_rb_u32x4 *)(g0 + 16 * rw) = *(const qzk_rb_u32x4 *)(ob + 16 * rw);
            qz_lds_sync();                                          /* row 0 has been read out */
            if ((total & 15) && lane < 4) ((uint32_t *)ob)[lane] = ((const uint32_t *)ob)[4 * R + (uint32_t)lane];
        }
    }
    S->obase = obase + Tb;
    return 0;
#undef QZK_OBI
}

/* n bytes from device memory straight to the output at obase (stored blocks, runs too long for a batch); the window must
 * have been flushed.  8 bytes a lane.  The caller orders later reads behind it (qz_wave_sync) and calls qzk_rb_skip. */
QZ_DEV void qzk_rb_direct(qzk_rb *S, const uint8_t *s, uint32_t n, int lane)
{
    uint8_t *d = S->o + S->obase;
    for (uint32_t i = 8 * (uint32_t)lane; i + 8 <= n; i += 512) qzk_rb_st64(d + i, qzk_rb_ld64(s + i));
    const uint32_t t = n & ~7u;
    if ((uint32_t)lane < (n & 7)) d[t + (uint32_t)lane] = s[t + (uint32_t)lane];
}
/* a match straight in the output (behind a flush and a qz_wave_sync): a byte per lane and step, period-wise */
QZ_DEV void qzk_rb_direct_match(qzk_rb *S, uint32_t D, uint32_t L, int lane)
{
    uint8_t *d = S->o + S->obase;
    const uint8_t *s = d - D;
    const uint32_t stp = 64u % D;
    uint32_t r = (uint32_t)lane % D;
    for (uint32_t i = (uint32_t)lane; i < L; i += 64) {
        d[i] = s[r];
        r += stp; if (r >= D) r -= D;
    }
}

#endif
