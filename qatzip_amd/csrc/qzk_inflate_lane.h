/*
 * qzk_inflate_lane.h — K3b: raw inflate of many segments in TWO PHASES, gfx950.
 *
 * Same contract as qzk_inflate_kernel (qzk_inflate.h: segment records, status codes,
 * count-only / through-flush flags) and the same place in the reference (zlib
 * inflate(), src/qatzip_sw.c:339), laid out for a call that holds thousands of
 * independent segments (a 2 GiB call = 32 768 chunks):
 *
 *   phase A  qzk_inflate_tok_kernel   ONE SEGMENT PER LANE.  Huffman decoding is
 *            bit-serial, so every lane decodes its own segment - but only decodes:
 *            literals go to a per-segment literal stream, matches become 8-byte
 *            sequence records (literal run, length, distance).  No match copies,
 *            no reads of its own output, one hot state (symbol decode) => short,
 *            barely divergent loop.  Root tables (9-bit literal/length, 7-bit
 *            distance) sit in LDS, 1.25 KiB per lane; the canonical ranges for the
 *            rare longer codes stay in a per-segment HBM record.
 *   phase B  qzk_lz_resolve_kernel    ONE SEGMENT PER WAVE.  64 sequences at a time:
 *            wave prefix sums give every sequence its output offset, the literals
 *            of the batch are scattered cooperatively (one byte per lane per step),
 *            then the matches are copied by their lanes in dependency order (a match
 *            is ready when its source lies below the first unfinished match).
 *
 * The single-phase ancestor of this file decoded AND copied in the lane: 6.6 M wave
 * instructions per 64 KiB segment, 47 % of the wave's time parked on its own
 * store->load round trips (profiles/).
 */
#ifndef QZK_INFLATE_LANE_H
#define QZK_INFLATE_LANE_H
#include "qzk_inflate.h"
#include "qzk_lz_batch.h"

#ifdef QZ_SIM
#define QZK_PIN(x) ((void)0)
#else
#define QZK_PIN(x) asm volatile("" : "+v"(x))       /* the value is needed HERE: a pending load is waited for at this point */
#endif

/* root tables (the per-symbol lookups) live in LDS; the canonical ranges for longer codes and the code lengths
 * stay in a per-segment HBM record */
#ifndef QZK_LLROOT
#define QZK_LLROOT 9
#endif
#ifndef QZK_LDROOT
#define QZK_LDROOT 7
#endif
/* Per lane, in LDS (QZK_LANE_ROOTSZ u16 = 1.25 KiB; the LDS is what bounds the lanes a CU holds, so every byte counts):
 *   [0, 1024)     literal/length root, 2^9 x u16: sym << 4 | code length, 0 = the code is longer than the root
 *   [1024, 1280)  the "side" region, 256 bytes:
 *       distance root, 2^QZK_LDROOT x u8: sym << 3 | code length (a distance symbol is < 30 and a root code <= 7 bits), then
 *       the LONG-LITERAL POOL, QZK_LPOOL_N x u8: the literal/length symbols whose codes are longer than the root, in
 *       canonical order (shortest codes = likeliest symbols first), as many as fit; 0xff = "not a literal below 255 -
 *       ask the segment's sorted list in memory".  On mixed data (text next to bytes of every value) a block has ~220
 *       such symbols carrying up to a fifth of its symbols (tools/inflate_longcodes.py), and fetching each of them
 *       from memory stood on every lane's critical path.
 * While a dynamic block's header is read, the side region holds the code-length code's root (2^7 x u16).
 * The side region's last QZK_SIDE_SPARE bytes belong to the kernel (round 6): phase A with four lanes a segment keeps its
 * group's END_BLOCK bookkeeping there between two headers - its sixteen rows of root tables are 20 480 bytes, an eighth of a
 * CU's LDS, and 448 bytes of arrays of its own on top were the eighth wave of every CU (profiles/r6_phaseA_order.txt). */
#define QZK_CLROOT 7
#define QZK_SIDE_BYTES 256
#define QZK_SIDE_SPARE 28
#define QZK_LPOOL_N (QZK_SIDE_BYTES - (1 << QZK_LDROOT) - QZK_SIDE_SPARE)
#define QZK_LANE_ROOTSZ ((1 << QZK_LLROOT) + QZK_SIDE_BYTES / 2)
#define QZK_DROOT8(droot) ((uint8_t *)(droot))
#define QZK_LPOOL(droot) ((uint8_t *)(droot) + (1 << QZK_LDROOT))

#ifdef QZ_SIM
QZ_DEV uint32_t qzk_rev15(uint32_t v) { return qzk_rev(v & 0x7fffu, 15); }
#else
QZ_DEV uint32_t qzk_rev15(uint32_t v) { return __builtin_bitreverse32(v) >> 17; }      /* the low 15 bits, reversed */
#endif

typedef struct {
    uint16_t lsorted[288], dsorted[32];
    uint16_t lcount[16], lfirst[16], lindex[16];
    uint16_t dcount[16], dfirst[16], dindex[16];
    uint8_t lens[320];
} qzk_inf_tab;

/* serial (per-lane) canonical table build; returns 0 ok, 1 incomplete, -1 over-subscribed */
/* FMT8: the root is the distance root (u8 entries, sym << 3 | len); pool: where the long literal/length symbols go */
template <bool FMT8 = false>
QZ_DEV int qzk_lane_build(const uint8_t *lens, int n, uint16_t *root, int rootbits, uint16_t *sorted,
                          uint16_t *count, uint16_t *first, uint16_t *index, int *maxlen_out, uint8_t *pool = 0)
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
    if (FMT8) { for (int i = 0; i < (1 << rootbits); i += 4) *(uint32_t *)((uint8_t *)root + i) = 0; }
    else for (int i = 0; i < (1 << rootbits); i += 2) *(uint32_t *)(root + i) = 0;
    if (pool) for (int i = 0; i < QZK_LPOOL_N; i += 4) *(uint32_t *)(pool + i) = 0xffffffffu;
    const uint32_t long_base = rootbits < 15 ? index[rootbits + 1] : off;    /* symbols with root codes come first in the sorted list */
    uint16_t next[16];
    for (int l = 1; l <= 15; l++) next[l] = 0;
    for (int i = 0; i < n; i++) {
        const int l = lens[i];
        if (!l) continue;
        const uint32_t rank = next[l]++;
        sorted[index[l] + rank] = (uint16_t)i;
        if (l <= rootbits) {
            const uint32_t r = qzk_rev(first[l] + rank, l);
            if (FMT8) { for (uint32_t f = r; f < (1u << rootbits); f += 1u << l) ((uint8_t *)root)[f] = (uint8_t)((i << 3) | l); }
            else for (uint32_t f = r; f < (1u << rootbits); f += 1u << l) root[f] = (uint16_t)((i << 4) | l);
        } else if (pool) {
            const uint32_t pr = index[l] + rank - long_base;
            if (pr < (uint32_t)QZK_LPOOL_N && i < 255) pool[pr] = (uint8_t)i;
        }
    }
    *maxlen_out = maxlen;
    return left > 0 ? 1 : 0;
}

/* per-lane bit reader.  The next input word is always already in flight (pw): a refill consumes it and issues the
 * load for the one after, so the load's latency is hidden behind the symbols decoded in between - and, vector
 * memory operations completing in order, it is issued ahead of this trip's stores instead of queueing behind them */
typedef struct { const uint8_t *p; uint32_t pos, end; uint64_t bb; int bc; uint32_t pw; bool pv; } qzk_lbits;

QZ_DEV void qzk_lseek(qzk_lbits *b, uint32_t pos)
{
    b->pos = pos; b->bb = 0; b->bc = 0;
    b->pv = pos + 4 <= b->end;
    b->pw = b->pv ? qz_ld32(b->p + pos) : 0;
}

QZ_DEV void qzk_lrefill(qzk_lbits *b)
{
    if (b->bc <= 32) {
        if (b->pv) {
            b->bb |= (uint64_t)b->pw << b->bc; b->pos += 4; b->bc += 32;
            b->pv = b->pos + 4 <= b->end;
            if (b->pv) b->pw = qz_ld32(b->p + b->pos);
        } else while (b->bc <= 56 && b->pos < b->end) { b->bb |= (uint64_t)b->p[b->pos++] << b->bc; b->bc += 8; }
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

typedef struct __attribute__((packed, aligned(1))) { uint64_t v; } qz_u64u;
QZ_DEV uint64_t qzk_ld64u(const uint8_t *p) { return ((const qz_u64u *)p)->v; }
QZ_DEV void qzk_st64u(uint8_t *p, uint64_t v) { ((qz_u64u *)p)->v = v; }

/* phase A -> phase B hand-over, per segment: literal bytes + sequence records */
typedef struct { uint32_t litrun; uint16_t mlen; uint16_t dm1; } qzk_seq;      /* mlen 0: literals only (tail) */
/* a sub-stream is ONE region of the scratch (round 6): its literals grow up from the region's first byte, its sequence
 * records DOWN from its end - record i lies at seqs[seq_off - 1 - i] - so that literal-heavy and match-heavy streams share
 * the same room (a byte of output is a literal or a third of a match at most: text needs 1.4 bytes of scratch per byte of
 * output, random data 1.0, only "a match every three bytes" the 2.67 that separate worst-case arrays had to hold both of) */
typedef struct { uint64_t lit_off; uint64_t seq_off; } qzk_tokseg;             /* lits + lit_off = the region's first byte; seqs + seq_off = its end */
/* what phase B stitches together for one segment: pieces of sub-streams, in output order.  The serial phase A
 * leaves one piece (the whole stream); the speculative one (qzk_inflate_spec.h) up to one per sub-decoder. */
#define QZK_SPEC_MAXK 8
#define QZK_INF_ESPEC (-6)         /* speculative phase A could not finish this segment: decode it serially */
#define QZK_PIECE_RAW 0xffffffffu  /* qzk_chain_el.sub: not a sub-stream but seq_count stored bytes at input offset seq_first */
typedef struct { uint32_t sub, seq_first, seq_count, lit_first, lrun_skip; } qzk_chain_el;
#ifndef QZK_CHAIN_MAXEL
#define QZK_CHAIN_MAXEL 640       /* pieces per segment (K per Huffman block of the segment, and per round that continues one): a 512 KB
                                   * segment holds ~20 blocks, sixteen lanes each (round 4: 160 - such segments came back through the
                                   * serial kernel, 1.5 GB/s); 12.5 KiB of device memory per segment */
#endif
typedef struct { uint32_t nel, pad; qzk_chain_el el[QZK_CHAIN_MAXEL]; } qzk_chain;
/* scratch a segment needs: literals <= out_cap (+ staging slack), sequences <= out_cap / 3 (+ tail) */
#define QZK_TOK_LITCAP(out_cap) ((((uint64_t)(out_cap) + 31) & ~(uint64_t)31) + 32)
#define QZK_TOK_SEQCAP(out_cap) (((uint64_t)(out_cap) / 3 + 3) & ~(uint64_t)1)      /* even: sequences leave in aligned pairs */
/* the region of a sub-stream that must hold a whole segment whatever it is made of (the one-lane kernel's): bytes, a multiple of 32 */
#define QZK_TOK_REGION(out_cap) (QZK_TOK_LITCAP(out_cap) + 8 * QZK_TOK_SEQCAP(out_cap))

enum { QZK_LS_HDR = 0, QZK_LS_SYM, QZK_LS_RAW, QZK_LS_DONE };
#ifndef QZK_LIT_RUN
#define QZK_LIT_RUN 4              /* literals one trip of the serial phase A may take (six, with the staging widened to match, measured 1.5 % slower) */
#endif

/* slow half of a symbol decode: the root entry was empty (code longer than the root) */
QZ_DEV int qzk_ldecode_long(qzk_lbits *b, int rootbits, const uint16_t *sorted, const uint16_t *count,
                            const uint16_t *first, const uint16_t *index, int maxlen)
{
    uint32_t code = qzk_rev((uint32_t)b->bb & ((1u << rootbits) - 1), rootbits);
    uint64_t bits = b->bb >> rootbits;
    int sym = -1;
    for (int l = rootbits + 1; l <= maxlen && sym < 0 && l <= b->bc; l++) {
        code = (code << 1) | (uint32_t)(bits & 1); bits >>= 1;
        const uint32_t c = count[l], f = first[l];
        if (c && code >= f && code - f < c) { QZK_DROP(b, l); sym = sorted[index[l] + code - f]; }
    }
    return sym;
}

/* The codes longer than the root table, without a loop and without memory.  Canonical codes read MSB first and
 * left-justified to 15 bits are ordered by length: the codes of length l fill [limit(l - 1), limit(l)) with
 * limit(l) = (first[l] + count[l]) << (15 - l).  So the length of the code in the next 15 bits V is ROOT + 1 + the number of
 * limits at or below V, and its index in the sorted symbol list V >> (15 - l) plus index[l] - first[l].  A lane keeps, for
 * the N = 15 - ROOT lengths above the root, one word each: limit << 16 | (index - first) & 0xffff - comparing
 * V << 16 | 0xffff against the word is the comparison of V against the limit - so a long code costs a compare, an add with
 * carry and a select per length (round 3 tested the N intervals one by one: three times the instructions, on a path some
 * lane of a wave takes in nearly every trip).  V at or above the last limit is no code at all (incomplete sets). */
#define QZK_LT_WORDS(ROOT) (15 - (ROOT))
#define QZK_LR_WORDS QZK_LT_WORDS(QZK_LLROOT)
#define QZK_DR_WORDS QZK_LT_WORDS(QZK_LDROOT)
template <int ROOT>
QZ_DEV void qzk_longtab_load_t(uint32_t *LR, const uint16_t *first_, const uint16_t *count_, const uint16_t *index_, int maxlen)
{
    constexpr int N = 15 - ROOT;
    (void)maxlen;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int l = ROOT + 1 + k;
        const uint32_t first = first_[l], limit = (first + count_[l]) << (15 - l);
        LR[k] = limit << 16 | (((uint32_t)index_[l] - first) & 0xffffu);
    }
}
/* (length, index in the sorted list) of the long code at the head of bb; length 0: no such code */
template <int ROOT>
QZ_DEV void qzk_long_interval(uint64_t bb, const uint32_t *LR, uint32_t *sel_l, uint32_t *sel_i)
{
    constexpr int N = 15 - ROOT;
    const uint32_t V = qzk_rev15((uint32_t)bb), VH = V << 16 | 0xffffu;
    uint32_t n = 0, w = LR[0];
#pragma unroll
    for (int k = 0; k < N - 1; k++) { const bool ge = VH >= LR[k]; n += ge ? 1u : 0u; w = ge ? LR[k + 1] : w; }
    const uint32_t l = (uint32_t)ROOT + 1u + n;
    *sel_l = VH >= LR[N - 1] ? 0u : l;
    *sel_i = ((V >> (15u - l)) + w) & 0xffffu;
}
/* CHECKBC: the reader may hold fewer valid bits than the code is long (the careful reader at the end of the input; the hot
 * loop's trips start with >= 56) */
template <int ROOT, bool CHECKBC = true>
QZ_DEV int qzk_ldecode_long_reg_t(qzk_lbits *b, const uint32_t *LR, const uint16_t *sorted, int maxlen)
{
    (void)maxlen;
    uint32_t sel_l, sel_i;
    qzk_long_interval<ROOT>(b->bb, LR, &sel_l, &sel_i);
    if (CHECKBC && (int)sel_l > b->bc) sel_l = 0;
    int sym = -1;
    if (sel_l) { QZK_DROP(b, sel_l); sym = sorted[sel_i]; }
    return sym;
}
QZ_DEV void qzk_longtab_load(uint32_t *LR, uint32_t *DR, const qzk_inf_tab *T, int lmax, int dmax)
{
    qzk_longtab_load_t<QZK_LLROOT>(LR, T->lfirst, T->lcount, T->lindex, lmax);
    qzk_longtab_load_t<QZK_LDROOT>(DR, T->dfirst, T->dcount, T->dindex, dmax);
}

/* per-lane decode state that the cold paths (block header, stored run) share with the kernel */
typedef struct {
    qzk_lbits b;
    uint32_t op, nblocks, last, clen, rpos, out_cap;
    uint32_t lbase;                     /* literal/length symbols with root codes = where the long ones begin in the sorted list */
    int lmax, dmax, status, state;
    bool through;
} qzk_lane_st;

/* ---- the block header and its tables, without memory on the way (round 4) ----
 * Round 3 kept a block's code lengths, counts and first codes in the segment's record in device memory and built the tables
 * from there: a store and a load per symbol, each a round trip, 1 M clocks per header with one lane of a group at work -
 * 13-21 % of a wave's time once the decode itself had become faster (profiles/r4_phaseA_timeline.txt).  Now everything a
 * header needs lives in the lane's own LDS and registers:
 *   - the code-length code (19 symbols, <= 7 bits) is decoded from registers: its limits by length, its symbols in
 *     canonical order five bits each - no table;
 *   - the code lengths go, four bits each, into the SIDE region (the literals 0..255: 128 bytes, where the distance root
 *     will stand afterwards) and, for the symbols from 256 on (30 literal/length symbols, every distance symbol), into
 *     four registers;
 *   - counts, first codes, list offsets and running ranks are sixteen 16-bit fields in four registers each (qzk_f16);
 *   - the literal/length root and the long-literal pool are filled straight from there (the pool stands in the side
 *     region's second half, which holds no lengths), then the distance root over the literals' lengths, which are done with.
 * Device memory is only WRITTEN (the sorted symbol lists and ranges the long-code paths of other lanes and kernels read). */
typedef struct { uint64_t a, b, c, d; } qzk_f16;                  /* sixteen 16-bit fields */
QZ_DEV uint32_t qzk_f16_get(const qzk_f16 *f, uint32_t l)
{
    const uint64_t w = l < 4 ? f->a : l < 8 ? f->b : l < 12 ? f->c : f->d;
    return (uint32_t)(w >> (16u * (l & 3u))) & 0xffffu;
}
QZ_DEV void qzk_f16_add(qzk_f16 *f, uint32_t l, uint32_t v)
{
    const uint64_t x = (uint64_t)v << (16u * (l & 3u));
    f->a += l < 4 ? x : 0; f->b += (l >= 4 && l < 8) ? x : 0; f->c += (l >= 8 && l < 12) ? x : 0; f->d += l >= 12 ? x : 0;
}
typedef struct { uint64_t w0, w1, w2, w3; } qzk_tail64;           /* the code lengths of symbols 256..319, four bits each */
QZ_DEV uint32_t qzk_tail_get(const qzk_tail64 *t, uint32_t i)      /* i = symbol - 256 */
{
    const uint64_t w = i < 16 ? t->w0 : i < 32 ? t->w1 : i < 48 ? t->w2 : t->w3;
    return (uint32_t)(w >> (4u * (i & 15u))) & 15u;
}
QZ_DEV void qzk_tail_put(qzk_tail64 *t, uint32_t i, uint32_t v)
{
    const uint64_t x = (uint64_t)v << (4u * (i & 15u));
    t->w0 |= i < 16 ? x : 0; t->w1 |= (i >= 16 && i < 32) ? x : 0; t->w2 |= (i >= 32 && i < 48) ? x : 0; t->w3 |= i >= 48 ? x : 0;
}
/* code length of literal/length-or-distance symbol i of the block being read (0..319) */
QZ_DEV uint32_t qzk_hlen(const uint8_t *side, const qzk_tail64 *t, uint32_t i)
{
    return i < 256 ? ((uint32_t)side[i >> 1] >> (4u * (i & 1u))) & 15u : qzk_tail_get(t, i - 256);
}

/* counts -> first codes and list offsets (canonical order); returns 0 ok, 1 incomplete, -1 over-subscribed */
QZ_DEV int qzk_canon(const qzk_f16 *cnt, qzk_f16 *first, qzk_f16 *index, int *maxlen_out, uint32_t *nsyms)
{
    int left = 1, maxlen = 0; uint32_t code = 0, off = 0;
    first->a = first->b = first->c = first->d = 0; index->a = index->b = index->c = index->d = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        const uint32_t c = qzk_f16_get(cnt, l);
        left <<= 1; left -= (int)c;
        if (left < 0) return -1;
        if (c) maxlen = (int)l;
        qzk_f16_add(first, l, code & 0xffffu); qzk_f16_add(index, l, off);
        code = (code + c) << 1; off += c;
    }
    *maxlen_out = maxlen; *nsyms = off;
    return left > 0 ? 1 : 0;
}
/* the ranges other lanes (and the careful paths) read from the segment's record: stores only */
QZ_DEV void qzk_ranges_out(const qzk_f16 *cnt, const qzk_f16 *first, const qzk_f16 *index, uint16_t *count_, uint16_t *first_, uint16_t *index_)
{
    for (uint32_t l = 0; l <= 15; l++) { count_[l] = (uint16_t)(l ? qzk_f16_get(cnt, l) : 0); first_[l] = (uint16_t)qzk_f16_get(first, l); index_[l] = (uint16_t)qzk_f16_get(index, l); }
}

/* literal/length tables of a block from its code lengths (side region + tail registers, or the fixed code's when side == NULL) */
QZ_DEV int qzk_build_litlen(const uint8_t *side, const qzk_tail64 *tl, uint32_t n, uint16_t *lroot, uint8_t *pool, qzk_inf_tab *T,
                            int *lmax, uint32_t *lbase)
{
#define QZK_LL_LEN(i) (side ? qzk_hlen(side, tl, (i)) : ((i) < 144 ? 8u : (i) < 256 ? 9u : (i) < 280 ? 7u : 8u))
    qzk_f16 cnt, first, index, next;
    cnt.a = cnt.b = cnt.c = cnt.d = 0; next.a = next.b = next.c = next.d = 0;
    for (uint32_t i = 0; i < n; i++) { const uint32_t l = QZK_LL_LEN(i); if (l) qzk_f16_add(&cnt, l, 1); }
    uint32_t nsyms;
    const int r = qzk_canon(&cnt, &first, &index, lmax, &nsyms);
    if (r < 0) return r;
    for (int i = 0; i < (1 << QZK_LLROOT); i += 2) *(uint32_t *)(lroot + i) = 0;
    for (int i = 0; i < QZK_LPOOL_N; i += 4) *(uint32_t *)(pool + i) = 0xffffffffu;
    const uint32_t long_base = QZK_LLROOT < 15 ? qzk_f16_get(&index, QZK_LLROOT + 1) : nsyms;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = QZK_LL_LEN(i);
        if (!l) continue;
        const uint32_t rank = qzk_f16_get(&next, l), at = qzk_f16_get(&index, l) + rank;
        qzk_f16_add(&next, l, 1);
        T->lsorted[at] = (uint16_t)i;
        if (l <= (uint32_t)QZK_LLROOT) {
            const uint32_t rv = qzk_rev(qzk_f16_get(&first, l) + rank, (int)l);
            for (uint32_t f = rv; f < (1u << QZK_LLROOT); f += 1u << l) lroot[f] = (uint16_t)((i << 4) | l);
        } else {
            const uint32_t pr = at - long_base;
            if (pr < (uint32_t)QZK_LPOOL_N && i < 255) pool[pr] = (uint8_t)i;
        }
    }
    qzk_ranges_out(&cnt, &first, &index, T->lcount, T->lfirst, T->lindex);
    *lbase = long_base;
    return r;
#undef QZK_LL_LEN
}
/* distance tables (root: u8 entries sym << 3 | len); the lengths of the ndist symbols from symbol `from` on, or all 5 (fixed) */
QZ_DEV int qzk_build_dist(const qzk_tail64 *tl, uint32_t from, uint32_t n, uint8_t *d8, qzk_inf_tab *T, int *dmax)
{
#define QZK_D_LEN(k) (tl ? qzk_tail_get(tl, (from) + (k) - 256u) : 5u)
    qzk_f16 cnt, first, index, next;
    cnt.a = cnt.b = cnt.c = cnt.d = 0; next.a = next.b = next.c = next.d = 0;
    for (uint32_t k = 0; k < n; k++) { const uint32_t l = QZK_D_LEN(k); if (l) qzk_f16_add(&cnt, l, 1); }
    uint32_t nsyms;
    const int r = qzk_canon(&cnt, &first, &index, dmax, &nsyms);
    if (r < 0) return r;
    for (int i = 0; i < (1 << QZK_LDROOT); i += 4) *(uint32_t *)(d8 + i) = 0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t l = QZK_D_LEN(k);
        if (!l) continue;
        const uint32_t rank = qzk_f16_get(&next, l);
        qzk_f16_add(&next, l, 1);
        T->dsorted[qzk_f16_get(&index, l) + rank] = (uint16_t)k;
        if (l <= (uint32_t)QZK_LDROOT) {
            const uint32_t rv = qzk_rev(qzk_f16_get(&first, l) + rank, (int)l);
            for (uint32_t f = rv; f < (1u << QZK_LDROOT); f += 1u << l) d8[f] = (uint8_t)((k << 3) | l);
        }
    }
    qzk_ranges_out(&cnt, &first, &index, T->dcount, T->dfirst, T->dindex);
    return r;
#undef QZK_D_LEN
}

/* one block header: stored -> RAW (or the end of the segment), fixed / dynamic -> tables built, SYM */
QZ_DEV void qzk_lane_header(qzk_lane_st *S, qzk_inf_tab *T, uint16_t *lroot, uint16_t *droot)
{
    qzk_lbits *b = &S->b;
    uint8_t *const side = (uint8_t *)droot;
    S->state = QZK_LS_DONE; S->status = QZK_INF_EDATA;       /* every early return below is a failure */
    qzk_lrefill(b);
    if (b->bc < 3) { S->status = QZK_INF_EIN; return; }
    S->last = QZK_GETBITS(b, 1); QZK_DROP(b, 1);
    const uint32_t type = QZK_GETBITS(b, 2); QZK_DROP(b, 2);
    S->nblocks++;
    if (type == 0) {
        QZK_DROP(b, b->bc & 7);
        qzk_lrefill(b);
        if (b->bc < 32) { S->status = QZK_INF_EIN; return; }
        const uint32_t len = QZK_GETBITS(b, 16); QZK_DROP(b, 16);
        const uint32_t nlen = QZK_GETBITS(b, 16); QZK_DROP(b, 16);
        if ((len ^ 0xffff) != nlen) return;
        const uint32_t ipos = b->pos - (uint32_t)(b->bc >> 3);
        if (ipos + len > b->end) { S->status = QZK_INF_EIN; return; }
        qzk_lseek(b, ipos + len);
        if (S->op + len > S->out_cap) { S->status = QZK_INF_EOUT; return; }
        if (len == 0) {
            if (S->last) S->status = QZK_INF_FINAL;
            else if (!S->through) S->status = QZK_INF_FLUSH;
            else S->state = QZK_LS_HDR;
            return;
        }
        S->clen = len; S->rpos = ipos; S->state = QZK_LS_RAW;
        return;
    }
    if (type == 3) return;
    if (type == 1) {
        qzk_build_litlen((const uint8_t *)0, (const qzk_tail64 *)0, 288, lroot, QZK_LPOOL(droot), T, &S->lmax, &S->lbase);
        qzk_build_dist((const qzk_tail64 *)0, 256, 30, QZK_DROOT8(droot), T, &S->dmax);
        S->state = QZK_LS_SYM;
        return;
    }
    qzk_lrefill(b);
    if (b->bc < 14) { S->status = QZK_INF_EIN; return; }
    const uint32_t nlen = QZK_GETBITS(b, 5) + 257; QZK_DROP(b, 5);
    const uint32_t ndist = QZK_GETBITS(b, 5) + 1; QZK_DROP(b, 5);
    const uint32_t ncode = QZK_GETBITS(b, 4) + 4; QZK_DROP(b, 4);
    if (nlen > 286 || ndist > 30) return;
    /* the code-length code: lengths of its 19 symbols (three bits each, in the format's order), canonical code in registers */
    uint64_t cl = 0;                                           /* symbol s: bits 3 s .. 3 s + 2 */
    for (uint32_t i = 0; i < ncode; i++) {
        qzk_lrefill(b);
        if (b->bc < 3) { S->status = QZK_INF_EIN; return; }
        const uint32_t v = QZK_GETBITS(b, 3); QZK_DROP(b, 3);
        const uint32_t ord = i < 6 ? ((16u | 17u << 5 | 18u << 10 | 0u << 15 | 8u << 20 | 7u << 25) >> (5 * i)) & 31
                           : i < 12 ? ((9u | 6u << 5 | 10u << 10 | 5u << 15 | 11u << 20 | 4u << 25) >> (5 * (i - 6))) & 31
                           : i < 18 ? ((12u | 3u << 5 | 13u << 10 | 2u << 15 | 14u << 20 | 1u << 25) >> (5 * (i - 12))) & 31 : 15u;
        cl |= (uint64_t)v << (3u * ord);
    }
    uint64_t climit = 0, cld = 0;                               /* per length l = 1..7: limit of its codes left-justified to 7 bits (8-bit
                                                                 * fields: 128 fits), list offset - first code (mod 256) */
    uint64_t cls0 = 0, cls1 = 0;                                /* the symbols in canonical order, five bits each, twelve a word */
    {
        uint32_t code = 0, off = 0; int left = 1;
        for (uint32_t l = 1; l <= 7; l++) {
            uint32_t c = 0;
            for (uint32_t sidx = 0; sidx < 19; sidx++) {
                if (((uint32_t)(cl >> (3u * sidx)) & 7u) == l) {
                    const uint32_t at = off + c;
                    if (at < 12) cls0 |= (uint64_t)sidx << (5u * at); else cls1 |= (uint64_t)sidx << (5u * (at - 12));
                    c++;
                }
            }
            left <<= 1; left -= (int)c;
            if (left < 0) return;
            climit |= (uint64_t)((code + c) << (7u - l)) << (8u * (l - 1));
            cld |= (uint64_t)((off - code) & 0xffu) << (8u * (l - 1));
            code = (code + c) << 1; off += c;
        }
        if (left != 0) return;                                  /* the code-length code must be complete */
    }
    qzk_tail64 tl; tl.w0 = tl.w1 = tl.w2 = tl.w3 = 0;
    {
        uint32_t i = 0, prev = 0, acc = 0;                      /* acc: the nibbles of the current group of eight literal symbols */
        while (i < nlen + ndist) {
            qzk_lrefill(b);
            if (b->bc < 7 && !(b->pos >= b->end)) return;
            /* one code-length symbol: the code's length is 1 + the number of limits at or below the next 7 bits read MSB first */
            const uint32_t V = qzk_rev((uint32_t)b->bb & 0x7fu, 7);
            uint32_t n7 = 0;
            for (uint32_t l = 1; l <= 6; l++) n7 += V >= ((uint32_t)(climit >> (8u * (l - 1))) & 0xffu) ? 1u : 0u;
            const uint32_t l7 = 1 + n7;
            if (V >= ((uint32_t)(climit >> 48) & 0xffu) || (int)l7 > b->bc) return;
            const uint32_t ci = ((V >> (7u - l7)) + ((uint32_t)(cld >> (8u * (l7 - 1))) & 0xffu)) & 0xffu;
            if (ci >= 19) return;
            const uint32_t sym = (uint32_t)((ci < 12 ? cls0 >> (5u * ci) : cls1 >> (5u * (ci - 12))) & 31u);
            QZK_DROP(b, l7);
            uint32_t rep = 1, val = sym;
            if (sym >= 16) {
                if (sym == 16) { if (i == 0 || b->bc < 2) return; val = prev; rep = 3 + QZK_GETBITS(b, 2); QZK_DROP(b, 2); }
                else if (sym == 17) { if (b->bc < 3) return; val = 0; rep = 3 + QZK_GETBITS(b, 3); QZK_DROP(b, 3); }
                else { if (b->bc < 7) return; val = 0; rep = 11 + QZK_GETBITS(b, 7); QZK_DROP(b, 7); }
                if (i + rep > nlen + ndist) return;
            }
            for (uint32_t k = 0; k < rep; k++, i++) {
                if (i < 256) {
                    acc |= val << (4u * (i & 7u));
                    if ((i & 7u) == 7u) { *(uint32_t *)(side + (i >> 3) * 4) = acc; acc = 0; }
                } else qzk_tail_put(&tl, i - 256, val);
            }
            prev = val;
        }
        /* (nlen >= 257: the literals' last group of eight was stored when symbol 255 came) */
    }
    if (qzk_tail_get(&tl, 0) == 0) return;                      /* END_BLOCK must have a code */
    int r = qzk_build_litlen(side, &tl, nlen, lroot, QZK_LPOOL(droot), T, &S->lmax, &S->lbase);
    if (r < 0 || (r > 0 && S->lmax != 1)) return;
    /* the distance root goes where the literals' lengths stood: they are done with */
    r = qzk_build_dist(&tl, nlen, ndist, QZK_DROOT8(droot), T, &S->dmax);
    if (r < 0 || (r > 0 && S->dmax > 1)) return;
    S->state = QZK_LS_SYM;
}

/* What phase A writes, per lane: literal bytes and 8-byte sequence records, each straight to its place in memory in the
 * trip that produced it - one dword store for the (at most QZK_LIT_RUN = 4) literals of a trip, whatever their number and
 * alignment (the bytes above them are overwritten by the next trip; the literal area has the slack), one 8-byte store for
 * a sequence.  Vector memory operations of a wave complete in order, so a store that is still on its way delays every
 * later load - but the only load of the hot loop is the input word of the NEXT trip, asked for a whole trip ahead and
 * issued before this trip's stores: by the time anything waits for it, the stores of the trip before are long done.
 * (Rounds 1-2 staged eight trips' worth in registers - 26 of them, two chains of selects per trip - and stored in rounds,
 * because their refill was a load on demand that queued behind every store; with the refill a trip ahead the staging only
 * cost instructions: profiles/r3_inflate_direct_stores.txt.) */
#define QZK_TOK_ROUND 8
#ifndef QZK_TOK_TRIPS
#define QZK_TOK_TRIPS 256           /* trips of the hot loop before the wave looks at its parked lanes again */
#endif
/* Round 4: a lane's tokens reach memory in ALIGNED 16-BYTE PIECES.  Round 3 stored what a trip decoded at once - a dword of
 * literals, an 8-byte sequence - and read its input 8 unaligned bytes a trip: three scattered accesses per lane and trip,
 * and the counters said that is what phase A was made of (TA busy 67 %, TD busy 85-91 % of the kernel's time, the vector
 * ALUs a fifth; profiles/r4_phaseA_pmc.txt) - the texture path serves about one scattered lane-address in four cycles,
 * whatever its size.  Now the literals of a trip go into a 16-byte register buffer that leaves as one dwordx4 store when it
 * is full (every ~5 trips), sequences leave in pairs, the input comes through a register window refilled 16 aligned bytes
 * at a time (qzk_win, every ~4.5 trips): 0.8 accesses per lane and trip instead of 3, paid for with ~60 ALU instructions.
 * What is staged: the literal bytes [lw & ~15, lw) in l0..l3 (l4: what ran over into the next piece), the sequence of an
 * odd count in q0.  qzk_tok_drain() stores what is complete; qzk_tok_finish() what is left. */
typedef uint32_t qzk_u32x4 __attribute__((vector_size(16)));
typedef struct {
    uint8_t *lp; qzk_seq *sq;           /* the region's first byte; its END: sequence i at sq[-1 - i] */
    uint32_t lrun, nseq;                /* literals since the last sequence, sequences so far */
    uint32_t lw;                        /* literal bytes so far */
    uint32_t l0, l1, l2, l3, l4;        /* the literal piece being filled */
    uint32_t lfull;                     /* 1: l0..l3 are a complete piece waiting to be stored (it begins at lw_piece) */
    uint64_t q0, q1;                    /* the sequence pair being filled */
    uint32_t qfull;                     /* 1: q0, q1 are a complete pair waiting to be stored (sequences nseq - 2, nseq - 1) */
    bool count_only;
} qzk_tok_out;
#define QZK_NLIT(O_) ((O_).lw)              /* literal bytes appended so far */

QZ_DEV void qzk_tok_init(qzk_tok_out *O, uint8_t *lp, qzk_seq *sq, bool count_only)
{
    O->lp = lp; O->sq = sq; O->count_only = count_only;
    O->lrun = 0; O->nseq = 0; O->lw = 0;
    O->l0 = O->l1 = O->l2 = O->l3 = O->l4 = 0; O->lfull = 0; O->q0 = O->q1 = 0; O->qfull = 0;
}
/* store the complete pieces (the literal piece ends at the last 16-byte boundary at or below lw; the pair at nseq) */
QZ_DEV void qzk_tok_drain(qzk_tok_out *O)
{
    if (O->lfull) {
        qzk_u32x4 v; v[0] = O->l0; v[1] = O->l1; v[2] = O->l2; v[3] = O->l3;
#ifndef QZK_X_NOMEM
        *(qzk_u32x4 *)(O->lp + ((O->lw & ~15u) - 16u)) = v;
#endif
        O->l0 = O->l4; O->l1 = 0; O->l2 = 0; O->l3 = 0; O->l4 = 0; O->lfull = 0;
    }
    if (O->qfull) {
        qzk_u32x4 v; v[0] = (uint32_t)O->q1; v[1] = (uint32_t)(O->q1 >> 32); v[2] = (uint32_t)O->q0; v[3] = (uint32_t)(O->q0 >> 32);     /* records nseq - 1, nseq - 2: downward */
#ifndef QZK_X_NOMEM
        *(qzk_u32x4 *)((uint64_t *)O->sq - O->nseq) = v;
#endif
        O->qfull = 0;
    }
}
/* append k (0..4) literals packed in v, lowest byte first, NOTHING above them; a piece that fills up waits in l0..l3 (lfull)
 * for the next drain - so at most one append between two drains may cross a 16-byte boundary: the callers drain first */
QZ_DEV void qzk_tok_lits4(qzk_tok_out *O, uint32_t v, uint32_t k)
{
    if (!O->count_only) {
        const uint32_t c = O->lw & 15u, di = c >> 2;
        const uint64_t t = (uint64_t)v << (8u * (c & 3u));
        const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
        O->l0 |= di == 0 ? lo : 0u;
        O->l1 |= di == 1 ? lo : di == 0 ? hi : 0u;
        O->l2 |= di == 2 ? lo : di == 1 ? hi : 0u;
        O->l3 |= di == 3 ? lo : di == 2 ? hi : 0u;
        O->l4 |= di == 3 ? hi : 0u;
        O->lfull = c + k >= 16u ? 1u : O->lfull;
        O->lw += k;
    }
    O->lrun += k;
}
/* the general form: k (1..8) literals, any state of the staging (the cold paths: the careful reader, stored bytes) */
QZ_DEV void qzk_tok_lits(qzk_tok_out *O, uint64_t v, uint32_t k)
{
    const uint32_t k0 = k < 4 ? k : 4;
    qzk_tok_drain(O);
    qzk_tok_lits4(O, (uint32_t)(k0 < 4 ? v & ((1ull << (8 * k0)) - 1) : v & 0xffffffffull), k0);
    if (k > 4) { qzk_tok_drain(O); qzk_tok_lits4(O, (uint32_t)((v >> 32) & ((k < 8 ? 1ull << (8 * (k - 4)) : 1ull << 32) - 1)), k - 4); }
}
QZ_DEV void qzk_tok_byte(qzk_tok_out *O, uint32_t byte) { qzk_tok_lits(O, byte, 1); }
/* append the low k (1..8) bytes of v */
QZ_DEV void qzk_tok_bytes(qzk_tok_out *O, uint64_t v, uint32_t k) { qzk_tok_lits(O, v, k); }
/* a sequence record (lrun literals, then a match; mlen 0: literals only), with the drain its pair may need first */
QZ_DEV void qzk_tok_seq_rec(qzk_tok_out *O, uint64_t rec)
{
    if (!O->count_only) {
        if (O->nseq & 1u) { O->q1 = rec; O->qfull = 1; } else O->q0 = rec;
    }
    O->nseq++; O->lrun = 0;
}
QZ_DEV void qzk_tok_seq(qzk_tok_out *O, uint32_t mlen, uint32_t dm1)
{
    qzk_tok_drain(O);
    qzk_tok_seq_rec(O, (uint64_t)O->lrun | (uint64_t)mlen << 32 | (uint64_t)dm1 << 48);
}
QZ_DEV void qzk_tok_round_flush(qzk_tok_out *O) { (void)O; }
/* everything staged goes to memory (the staging stays valid: more may be appended afterwards) */
QZ_DEV void qzk_tok_flush(qzk_tok_out *O)
{
    if (O->count_only) return;
    qzk_tok_drain(O);
    if (O->lw & 15u) { qzk_u32x4 v; v[0] = O->l0; v[1] = O->l1; v[2] = O->l2; v[3] = O->l3; *(qzk_u32x4 *)(O->lp + (O->lw & ~15u)) = v; }
    if (O->nseq & 1u) *((uint64_t *)O->sq - O->nseq) = O->q0;
}

QZ_DEV void qzk_tok_finish(qzk_tok_out *O)
{
    if (O->count_only) return;
    if (O->lrun) qzk_tok_seq(O, 0u, 0u);
    qzk_tok_flush(O);
}

/* The input side of the same idea: a lane's next 32..48 bytes of input in registers.  w0..w7 hold the 32 bytes at `base` (a
 * multiple of 16 from the lane's first aligned address), n0..n3 the 16 after them, asked for when the window last moved;
 * the 8 bytes at any position in [base, base + 16) are three of w0..w5 and two funnel shifts.  qzk_win_step() moves the
 * window when the position has left its first half: it reads n0..n3 (a load a whole window step old), and the load for the
 * next step goes out LAST - behind the drain's stores - so that the wait in front of the read never meets anything younger
 * than a trip.  (Named registers, not arrays: a select over an array becomes an indexed load from scratch.) */
typedef struct { uint32_t w0, w1, w2, w3, w4, w5, w6, w7, n0, n1, n2, n3; uint32_t base; } qzk_win;
QZ_DEV uint32_t qzk_win_edge(const uint8_t *p, int32_t q)       /* the dword at p + q, bytes before p read as zero */
{
    uint32_t v = 0;
    for (int j = 0; j < 4; j++) if (q + j >= 0) v |= (uint32_t)p[q + j] << (8 * j);
    return v;
}
QZ_DEV void qzk_win_init(qzk_win *W, const uint8_t *p, uint32_t pos)
{
    /* aligned to 16 in memory: (p + base) % 16 == 0, base <= pos < base + 16.  base may be "negative" (before p) by up to 15
     * bytes: the buffers start well inside their allocation's first page only by luck, so the first piece is read bytewise
     * where it would begin before p */
    const uint32_t mis = (uint32_t)((uintptr_t)p & 15u);
    const uint32_t base = ((pos + mis) & ~15u) - mis;           /* wraps below zero when pos + mis < 16 and mis != 0 */
    W->base = base;
    if ((int32_t)base < 0) {
        W->w0 = qzk_win_edge(p, (int32_t)base); W->w1 = qzk_win_edge(p, (int32_t)base + 4);
        W->w2 = qzk_win_edge(p, (int32_t)base + 8); W->w3 = qzk_win_edge(p, (int32_t)base + 12);
    } else { const qzk_u32x4 a = *(const qzk_u32x4 *)(p + base); W->w0 = a[0]; W->w1 = a[1]; W->w2 = a[2]; W->w3 = a[3]; }
    const qzk_u32x4 b = *(const qzk_u32x4 *)(p + (int32_t)base + 16), c = *(const qzk_u32x4 *)(p + (int32_t)base + 32);
    W->w4 = b[0]; W->w5 = b[1]; W->w6 = b[2]; W->w7 = b[3];
    W->n0 = c[0]; W->n1 = c[1]; W->n2 = c[2]; W->n3 = c[3];
}
/* move the window if pos has left its first half (returns whether: the caller issues the next load after its stores) */
QZ_DEV bool qzk_win_step(qzk_win *W, uint32_t pos)
{
    const bool mv = pos - W->base >= 16u;
    if (mv) {
        W->w0 = W->w4; W->w1 = W->w5; W->w2 = W->w6; W->w3 = W->w7;
        W->w4 = W->n0; W->w5 = W->n1; W->w6 = W->n2; W->w7 = W->n3;
        W->base += 16u;
    }
    return mv;
}
QZ_DEV void qzk_win_load(qzk_win *W, const uint8_t *p, bool mv)
{
#ifdef QZK_X_NOMEM         /* timing experiments only: no memory traffic in the loop (the decode then runs on what the window held) */
    (void)p; (void)mv;
#else
    if (mv) { const qzk_u32x4 c = *(const qzk_u32x4 *)(p + (int32_t)W->base + 32); W->n0 = c[0]; W->n1 = c[1]; W->n2 = c[2]; W->n3 = c[3]; }
#endif
}
/* the 8 bytes at pos, base <= pos < base + 16 */
QZ_DEV uint64_t qzk_win_get(const qzk_win *W, uint32_t pos)
{
    const uint32_t o = pos - W->base, d = o >> 2, sh = 8u * (o & 3u);
    /* (pinned: a select over values the compiler can trace to one struct becomes an indexed load from a scratch copy) */
    uint32_t x0 = W->w0, x1 = W->w1, x2 = W->w2, x3 = W->w3, x4 = W->w4, x5 = W->w5;
    QZK_PIN(x0); QZK_PIN(x1); QZK_PIN(x2); QZK_PIN(x3); QZK_PIN(x4); QZK_PIN(x5);
    const uint32_t a = d == 0 ? x0 : d == 1 ? x1 : d == 2 ? x2 : x3;
    const uint32_t b = d == 0 ? x1 : d == 1 ? x2 : d == 2 ? x3 : x4;
    const uint32_t c = d == 0 ? x2 : d == 1 ? x3 : d == 2 ? x4 : x5;
    const uint64_t ab = (uint64_t)a | (uint64_t)b << 32, bc = (uint64_t)b | (uint64_t)c << 32;
    return (uint64_t)(uint32_t)(ab >> sh) | (uint64_t)(uint32_t)(bc >> sh) << 32;
}

/* one symbol from an already refilled bit buffer; structured (no early exits) so that it compiles to predicated
 * straight-line code.  MIDREFILL: the caller's reader guarantees < 48 valid bits, refill before the distance code. */
template <bool MIDREFILL, bool PAIR>
QZ_DEV void qzk_lane_symbol(qzk_lane_st *S, qzk_tok_out *O, const qzk_inf_tab *T, const uint16_t *lroot,
                            const uint16_t *droot, uint64_t hist, const uint32_t *LR = 0, const uint32_t *DR = 0)
{
    qzk_lbits *b = &S->b;
    const uint32_t e = lroot[(uint32_t)b->bb & ((1u << QZK_LLROOT) - 1)];
    int sym;
    if (e != 0 && (!MIDREFILL || (int)(e & 15) <= b->bc)) { QZK_DROP(b, e & 15); sym = (int)(e >> 4); }     /* a trip of the hot loop starts with >= 56 bits */
    else if (e) sym = -1;
    else if (LR) sym = qzk_ldecode_long_reg_t<QZK_LLROOT, MIDREFILL>(b, LR, T->lsorted, S->lmax);
    else sym = qzk_ldecode_long(b, QZK_LLROOT, T->lsorted, T->lcount, T->lfirst, T->lindex, S->lmax);
    uint64_t lv = 0; uint32_t lk = 0;       /* literals of this trip (packed), their number */
    bool run = false;                       /* the trip may go on with literals out of the root table */
    if (sym < 256) {
        if (sym < 0) { S->status = b->pos >= b->end && b->bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA; S->state = QZK_LS_DONE; }
        else if (S->op >= S->out_cap) { S->status = QZK_INF_EOUT; S->state = QZK_LS_DONE; }
        else { lv = (uint64_t)sym; lk = 1; run = PAIR; }
    } else if (sym == 256) {
        if (S->last) { S->status = QZK_INF_FINAL; S->state = QZK_LS_DONE; } else S->state = QZK_LS_HDR;
    } else {
        sym -= 257;
        uint32_t xb = (sym < 8 || sym >= 28) ? 0u : (uint32_t)(sym - 4) >> 2;
        uint32_t len = sym < 8 ? 3u + (uint32_t)sym : sym >= 28 ? 258u : 3u + ((4u + ((uint32_t)sym & 3)) << xb);
        int err = sym >= 29 ? QZK_INF_EDATA : (int)xb > b->bc ? QZK_INF_EIN : 0;
        len += QZK_GETBITS(b, xb); QZK_DROP(b, xb);
        if (MIDREFILL) qzk_lrefill(b);
        const uint32_t de = QZK_DROOT8(droot)[(uint32_t)b->bb & ((1u << QZK_LDROOT) - 1)];
        int ds;
        if (de != 0 && (int)(de & 7) <= b->bc) { QZK_DROP(b, de & 7); ds = (int)(de >> 3); }
        else if (de) ds = -1;
        else if (DR) ds = qzk_ldecode_long_reg_t<QZK_LDROOT, MIDREFILL>(b, DR, T->dsorted, S->dmax);
        else ds = qzk_ldecode_long(b, QZK_LDROOT, T->dsorted, T->dcount, T->dfirst, T->dindex, S->dmax);
        if (!err && (ds < 0 || ds >= 30)) err = b->pos >= b->end && b->bc < 15 ? QZK_INF_EIN : QZK_INF_EDATA;
        if (ds < 0 || ds >= 30) ds = 0;
        xb = ds < 4 ? 0u : (uint32_t)(ds - 2) >> 1;
        uint32_t dist = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1)) << xb);
        if (!err && (int)xb > b->bc) err = QZK_INF_EIN;
        dist += QZK_GETBITS(b, xb); QZK_DROP(b, xb);
        if (!err && (uint64_t)dist > hist + S->op) err = QZK_INF_EHIST;   /* hist: bytes before the segment a match may reach */
        if (!err && S->op + len > S->out_cap) err = QZK_INF_EOUT;
        if (err) { S->status = err; S->state = QZK_LS_DONE; }
        else { qzk_tok_seq(O, len, dist - 1); S->op += len; run = PAIR; }
    }
    if (PAIR) {
        /* literals come in runs, and a match is usually followed by some: the trip goes on for up to QZK_LIT_RUN literals
         * in all (three after a match) while their codes sit in the 9-bit root table and the bits the trip started with
         * last (56; every step checks what is left).  Literal-heavy segments are phase A's critical path: the least
         * compressible segments have the most symbols, and a lane decodes its segment alone.  Not for the speculative
         * decoders, whose trips must start at every symbol boundary their neighbour may have published. */
        for (int extra = 0; extra < QZK_LIT_RUN - 1; extra++) {
            const uint32_t e2 = lroot[(uint32_t)b->bb & ((1u << QZK_LLROOT) - 1)];
            run = run && e2 != 0 && (e2 >> 4) < 256 && (int)(e2 & 15) <= b->bc && S->op + lk < S->out_cap;
            if (run) {
                QZK_DROP(b, e2 & 15);
                lv |= (uint64_t)(e2 >> 4) << (8 * lk); lk++;
            }
        }
    }
    if (lk) { qzk_tok_lits(O, lv, lk); S->op += lk; }
}


/* ------------------------------------------------------------------ the serial phase A's trip (round 4)
 * A lane's segment is ONE dependent chain - bits -> root lookup -> code length -> shift -> next lookup - and the phase
 * ends when the longest chain does, so a trip is written for that chain and nothing else:
 *   - the trip's stores are UNCONDITIONAL and come last: one literal store (however many literals, zero included: the
 *     bytes above them are overwritten by the next trip) and one sequence store (a trip without a sequence writes the
 *     slot the next sequence will overwrite).  A constant number of memory operations behind the next trip's input load
 *     lets the compiler wait for that load alone (s_waitcnt vmcnt(2)); with stores under branches it had to wait for
 *     vmcnt(0), i.e. for this trip's stores to reach the L2, at the top of every trip;
 *   - codes longer than the root are settled without memory where it matters: the six intervals of the lengths 10..15
 *     in registers give the length and the canonical index, and a long LITERAL is one more LDS read (the pool); the
 *     distance code's long half is registers only (intervals + the <= 30 symbols packed five bits each);
 *   - selects instead of branches on the common path; what is rare (a long code, the end of a block, an error) sits in
 *     blocks the wave skips when no lane needs them.
 * Same tokens as qzk_lane_symbol<false, true> (the careful reader and the speculative decoders keep using that one). */
typedef struct { uint64_t w0, w1, w2; } qzk_dsyms;       /* <= 30 distance symbols in canonical order, five bits each, twelve per word
                                                           * (named words, not an array: the compiler turns a select over an array
                                                           * into an indexed load from scratch) */
QZ_DEV uint64_t qzk_dsyms_word(const uint16_t *dsorted, uint32_t n, uint32_t k)
{
    uint64_t word = 0;
    for (uint32_t j = 0; j < 12; j++) { const uint32_t i = 12u * k + j; if (i < n) word |= (uint64_t)((uint32_t)dsorted[i] & 31u) << (5 * j); }
    return word;
}
QZ_DEV void qzk_dsyms_load(qzk_dsyms *DS, const uint16_t *dsorted, const uint16_t *dcount)
{
    uint32_t n = 0;
    for (int l = 1; l <= 15; l++) n += dcount[l];
    if (n > 30) n = 30;
    DS->w0 = qzk_dsyms_word(dsorted, n, 0); DS->w1 = qzk_dsyms_word(dsorted, n, 1); DS->w2 = qzk_dsyms_word(dsorted, n, 2);
}

template <int NX>      /* NX: literals a trip may take behind its first symbol (none when !allow: the speculative decoders near a mark) */
QZ_DEV void qzk_lane_trip(qzk_lane_st *S, qzk_tok_out *O, const qzk_inf_tab *T, const uint16_t *lroot, const uint16_t *droot,
                          uint64_t hist, const uint32_t *LR, const uint32_t *DR, const qzk_dsyms *DS, bool allow = true)
{
    uint64_t bb = S->b.bb; int bc = S->b.bc;                    /* >= 56 valid bits: a whole symbol with everything it drags along */
    const uint8_t *const d8 = QZK_DROOT8(droot), *const pool = QZK_LPOOL(droot);
    const uint32_t e = lroot[(uint32_t)bb & ((1u << QZK_LLROOT) - 1)];
    uint32_t l = e & 15; int sym = (int)(e >> 4);
    if (e == 0) {                                               /* longer than the root */
        uint32_t sl, si;
        qzk_long_interval<QZK_LLROOT>(bb, LR, &sl, &si);
        const uint32_t pr = si - S->lbase;
        const uint32_t pv = pool[pr < (uint32_t)QZK_LPOOL_N ? pr : 0u];
        sym = -1; l = sl;
        if (sl != 0) {
            if (pr < (uint32_t)QZK_LPOOL_N && pv != 0xffu) sym = (int)pv;
            else { sym = T->lsorted[si]; QZK_PIN(sym); }       /* a long length code, END_BLOCK, a literal the pool has no room for;
                                                                 * pinned: the wait for this load belongs in here, not where the paths meet.
                                                                 * (Asking for it and sitting the trip out instead - the symbol at hand when
                                                                 * the next trip begins - was measured: 4 % slower, the extra trips cost
                                                                 * more than the wait) */
        }
    }
    bb >>= l; bc -= (int)l;
    uint32_t room = S->out_cap - S->op;                         /* bytes the segment may still produce */
    uint32_t lv = 0, lk = 0;                                    /* the trip's literals (packed, lowest first), their number */
    uint64_t lv_hi = 0;
    uint64_t rec = 0; uint32_t nrec = 0;                        /* the trip's sequence, if it has one */
    bool run = false;
    const bool is_lit = (uint32_t)sym < 256u;
    if (is_lit && room != 0) { lv = (uint32_t)sym; lk = 1; run = allow; }
    if (sym > 256) {
        const int ls = sym - 257;
        uint32_t xb = (ls < 8 || ls >= 28) ? 0u : (uint32_t)(ls - 4) >> 2;
        uint32_t len = ls < 8 ? 3u + (uint32_t)ls : ls >= 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3)) << xb);
        len += (uint32_t)bb & ((1u << xb) - 1); bb >>= xb; bc -= (int)xb;
        const uint32_t de = d8[(uint32_t)bb & ((1u << QZK_LDROOT) - 1)];
        uint32_t dl = de & 7; int ds = (int)(de >> 3);
        if (de == 0) {
            uint32_t sl, si;
            qzk_long_interval<QZK_LDROOT>(bb, DR, &sl, &si);
            const uint32_t w = (si * 43u) >> 9, sh = 5u * (si - 12u * w);
            const uint64_t word = w == 0 ? DS->w0 : w == 1 ? DS->w1 : DS->w2;
            ds = sl != 0 && si < 30 ? (int)((uint32_t)(word >> sh) & 31u) : -1; dl = sl;
        }
        bb >>= dl; bc -= (int)dl;
        const bool dbad = ds < 0 || ds >= 30;
        if (dbad) ds = 0;
        xb = ds < 4 ? 0u : (uint32_t)(ds - 2) >> 1;
        uint32_t dist = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + ((uint32_t)ds & 1)) << xb);
        dist += (uint32_t)bb & ((1u << xb) - 1); bb >>= xb; bc -= (int)xb;
        const int err = ls >= 29 || dbad ? QZK_INF_EDATA : (uint64_t)dist > hist + S->op ? QZK_INF_EHIST : len > room ? QZK_INF_EOUT : 0;
        if (err) { S->status = err; S->state = QZK_LS_DONE; }
        else {
            rec = (uint64_t)O->lrun | (uint64_t)len << 32 | (uint64_t)(dist - 1) << 48; nrec = 1;
            S->op += len; room -= len; run = allow;
        }
    } else if (!(is_lit && room != 0)) {                        /* the end of the block, or of the decode */
        if (sym == 256) { if (S->last) { S->status = QZK_INF_FINAL; S->state = QZK_LS_DONE; } else S->state = QZK_LS_HDR; }
        else { S->status = sym < 0 ? QZK_INF_EDATA : QZK_INF_EOUT; S->state = QZK_LS_DONE; }
    }
    /* literals come in runs, and a match is usually followed by some: up to NX more while their codes sit in the root
     * table, the bits the trip started with last and the segment has room */
#pragma unroll
    for (int x = 0; x < NX; x++) {
        const uint32_t e2 = lroot[(uint32_t)bb & ((1u << QZK_LLROOT) - 1)];
        const uint32_t l2 = e2 & 15;
        run = run && (e2 - 1u) < 4095u && (int)l2 <= bc && lk < room;
        const uint32_t take = run ? l2 : 0u;
        bb >>= take; bc -= (int)take;
        lv |= run ? (e2 >> 4) << (8 * lk) : 0u;
        lk += run ? 1u : 0u;
    }
    /* the trip's tokens go into the staging (the caller drained it before the trip: one append may complete a piece) */
    static_assert(NX <= 3, "a trip's literals are one dword of the staging");
    (void)lv_hi;
    if (nrec) qzk_tok_seq_rec(O, rec);
    qzk_tok_lits4(O, lv, lk);
    S->op += lk;
    S->b.bb = bb; S->b.bc = bc;
}

/* LPW = segments (active lanes) per single-wave workgroup: LPW * 1.25 KiB of LDS (16 -> eight workgroups per CU) */
/* OCC = waves per SIMD the register budget is cut for (LPW 16: the LDS admits 8 waves per CU = 2 per SIMD; LPW 8: 16 = 4) */
template <int LPW, int OCC = 2>
QZ_KERNEL_OCC(64, OCC) qzk_inflate_tok_kernel(const uint8_t *comp, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs,
                                 qzk_inf_tab *tabs, const qzk_tokseg *ts, uint8_t *lits, qzk_seq *seqs,
                                 qzk_chain *chains, const uint32_t *order = 0 /* or: the launch covers `count` segments picked by index */,
                                 uint32_t count = 0, uint32_t ts_stride = 1 /* sub-streams per segment in ts: this kernel fills the first */)
{
    QZ_LDS uint16_t roots[LPW][QZK_LANE_ROOTSZ];
    const uint32_t widx = blockIdx.x * LPW + threadIdx.x;
    if (widx >= (order ? count : nsegs)) return;
    const uint32_t sidx = order ? order[widx] : widx;
    if (sidx >= nsegs) return;
    const qzk_infseg sg = segs[sidx];
    qzk_inf_tab *T = tabs + sidx;
    uint16_t *const lroot = roots[threadIdx.x], *const droot = lroot + (1 << QZK_LLROOT);
    qzk_lane_st S;
    S.b.p = comp + sg.in_off; S.b.end = sg.in_len; qzk_lseek(&S.b, 0);
    S.op = 0; S.nblocks = 0; S.last = 0; S.clen = 0; S.rpos = 0; S.out_cap = sg.out_cap;
    S.lmax = 0; S.dmax = 0; S.lbase = 0; S.status = QZK_INF_EDATA; S.state = QZK_LS_HDR;
    S.through = sg.flags & QZK_INF_THROUGH_FLUSH;
    qzk_tok_out O;
    {
        const bool co = sg.flags & QZK_INF_COUNT_ONLY;
        qzk_tok_init(&O, co ? lits : lits + ts[(uint64_t)sidx * ts_stride].lit_off, co ? seqs : seqs + ts[(uint64_t)sidx * ts_stride].seq_off, co);
    }
    /* the pieces phase B puts the segment together from: runs of sequences, and stored blocks as they lie in the input */
    qzk_chain *const C = chains + sidx;
    uint32_t nel = 0, piece_seq0 = 0, piece_lit0 = 0;
#ifdef QZK_INF_PROF
    uint32_t prof_trips = 0; const uint64_t prof_t0 = __builtin_readcyclecounter();
#endif
    uint32_t LR[QZK_LR_WORDS], DR[QZK_DR_WORDS];    /* the long literal/length and distance codes of the current block (registers) */
    for (int i = 0; i < QZK_LR_WORDS; i++) LR[i] = 0;
    for (int i = 0; i < QZK_DR_WORDS; i++) DR[i] = 0;
    qzk_dsyms DS; DS.w0 = DS.w1 = DS.w2 = 0;        /* the distance symbols of the current block */

    while (S.state != QZK_LS_DONE) {
        qzk_lbits *b = &S.b;
        if (S.state == QZK_LS_SYM && b->pos + 64 <= b->end && !O.count_only) {
            /* ---- the hot loop.  Every trip starts with the lane's memory traffic, such as it is: the input window moves on
             * when the read position has left its first half (reading the 16 bytes asked for at the last move), the token
             * pieces that are complete are stored, and the load for the window's next move goes out last - whatever a trip
             * waits for is a whole trip old.  Then a branch-free refill: the 8 bytes at the read position come out of the
             * window, are ORed in above the valid bits, the position steps over the bytes that fitted; >= 56 valid bits per
             * trip is a whole symbol (15 + 5 + 15 + 13).  Bounded (QZK_TOK_TRIPS trips) so that lanes parked in a cold
             * state - the next block's header - get their turn. ---- */
            b->pos -= (uint32_t)(b->bc >> 3); b->bc &= 7; b->bb &= (1ull << b->bc) - 1;     /* hand whole bytes back */
            qzk_win W; qzk_win_init(&W, b->p, b->pos);
            const uint64_t hist = S.through ? sg.out_off : 0;
            for (int trip = 0; trip < QZK_TOK_TRIPS && S.state == QZK_LS_SYM && b->pos + 64 <= b->end; trip++) {
                const bool mv = qzk_win_step(&W, b->pos);
                qzk_tok_drain(&O);
                qzk_win_load(&W, b->p, mv);
                b->bb |= qzk_win_get(&W, b->pos) << b->bc;
                b->pos += (uint32_t)(63 - b->bc) >> 3; b->bc |= 56;
                qzk_lane_trip<QZK_LIT_RUN - 1>(&S, &O, T, lroot, droot, hist, LR, DR, &DS);
#ifdef QZK_INF_PROF
                prof_trips++;
#endif
            }
            b->pos -= (uint32_t)(b->bc >> 3); b->bc &= 7; b->bb &= (1ull << b->bc) - 1;
            const uint32_t keep = (uint32_t)b->bc; const uint64_t low = b->bb;
            qzk_lseek(b, b->pos);                      /* back to the careful reader: its prefetch word */
            b->bb = low; b->bc = (int)keep;
        } else if (S.state == QZK_LS_SYM) {
            /* last bytes of the input: the careful reader */
            for (int round = 0; round < 8 && S.state == QZK_LS_SYM; round++) {
                for (int trip = 0; trip < QZK_TOK_ROUND && S.state == QZK_LS_SYM; trip++) {
                    qzk_lrefill(b);
                    qzk_lane_symbol<true, true>(&S, &O, T, lroot, droot, S.through ? sg.out_off : 0, LR, DR);
                }
                qzk_tok_round_flush(&O);
            }
        } else if (S.state == QZK_LS_HDR) {
            qzk_lane_header(&S, T, lroot, droot);
            if (S.state == QZK_LS_SYM) { qzk_longtab_load(LR, DR, T, S.lmax, S.dmax); qzk_dsyms_load(&DS, T->dsorted, T->dcount); }
        }
        else if (S.state == QZK_LS_RAW) {
            if (O.count_only) { S.op += S.clen; S.rpos += S.clen; S.clen = 0; }         /* nothing to move */
            else if (nel + 3 <= QZK_CHAIN_MAXEL) {
                /* a stored block never enters the literal stream: phase B copies it straight from the input.  The
                 * sequences so far become a piece of their own (their pending literals end it) */
                if (O.lrun) qzk_tok_seq(&O, 0u, 0u);
                /* (round 2 staged sequences in registers and lost one here - literals + stored block + eight matches;
                 * tests/test_sim_kernels.py keeps the case.  Since round 3 a sequence is in memory when it is appended.) */
                if (O.nseq > piece_seq0) {
                    qzk_chain_el e; e.sub = 0; e.seq_first = piece_seq0; e.seq_count = O.nseq - piece_seq0; e.lit_first = piece_lit0; e.lrun_skip = 0;
                    C->el[nel++] = e;
                }
                qzk_chain_el r; r.sub = QZK_PIECE_RAW; r.seq_first = S.rpos; r.seq_count = S.clen; r.lit_first = 0; r.lrun_skip = 0;
                C->el[nel++] = r;
                piece_seq0 = O.nseq; piece_lit0 = QZK_NLIT(O);
                S.op += S.clen; S.rpos += S.clen; S.clen = 0;
            } else {
                /* (more stored blocks than a chain holds pieces) its bytes join the literal stream, <= 8 per trip */
                for (int trip = 0; trip < 64 && S.clen; trip++) {
                    const uint32_t k = S.clen < 8 ? S.clen : 8;
                    uint64_t v = 0;
                    if (S.rpos + 8 <= b->end) v = qzk_ld64u(b->p + S.rpos);
                    else for (uint32_t i = 0; i < k; i++) v |= (uint64_t)b->p[S.rpos + i] << (8 * i);
                    qzk_tok_bytes(&O, v, k);
                    qzk_tok_round_flush(&O);
                    S.op += k; S.rpos += k; S.clen -= k;
                }
            }
            if (!S.clen) { if (S.last) { S.status = QZK_INF_FINAL; S.state = QZK_LS_DONE; } else S.state = QZK_LS_HDR; }
        }
    }
    qzk_tok_finish(&O);
    if (!O.count_only) {
        if (O.nseq > piece_seq0 || nel == 0) {
            qzk_chain_el e; e.sub = 0; e.seq_first = piece_seq0; e.seq_count = O.nseq - piece_seq0; e.lit_first = piece_lit0; e.lrun_skip = 0;
            C->el[nel++] = e;
        }
        C->nel = nel; C->pad = 0;
    }
    qzk_infres r;
    r.status = S.status; r.out_len = S.op; r.nblocks = S.nblocks;
    r.in_used = S.b.pos - (uint32_t)(S.b.bc >> 3);
#ifdef QZK_INF_PROF                     /* profiling builds only: trips of the hot loop / shader cycles of this lane */
    r.nblocks = prof_trips; r.in_used = (uint32_t)((__builtin_readcyclecounter() - prof_t0) >> 6);
#endif
    res[sidx] = r;
}

/* ------------------------------------------------------------------ phase B */
QZ_DEV uint32_t qzk_wave_scan_incl(uint32_t v, int lane) { (void)lane; return qz_wave_incl_scan(v); }

/* phase B orders its direct writes (stored blocks, runs too long for a batch) before whatever reads them back */
#define QZK_RSYNC() qz_wave_sync()

#ifdef QZK_SPEC_PROF
__device__ unsigned long long qzk_stamp_b[2];
#endif
/* one wave per segment; QZK_RES_WAVES segments per workgroup (one: single-wave workgroups).  ts_stride sub-streams per segment (1 after the serial
 * phase A, K after the speculative one); the chain says which pieces of which sub-streams make up the segment.  The
 * output-dependent checks (capacity, history) live here because a speculative sub-decoder does not know its offset.
 * The copying itself is qzk_lz_batch.h (round 5): batches of up to 64 sequences put together in an LDS window, whole 16-byte
 * rows out, one exposed memory round trip per batch. */
#ifndef QZK_RES_WAVES
#define QZK_RES_WAVES 1            /* segments (waves) per workgroup: 1 / 2 / 4 / 8 -> 11.6 / 11.9 / 13.1 / 15.0 ms per 2 GiB call (round 4) - a
                                    * workgroup leaves with its slowest segment, and segments differ */
#endif
#ifndef QZK_RES_OCC
#define QZK_RES_OCC 8              /* waves per SIMD the register budget is cut for: ~57 VGPRs, nothing in scratch (the segment index is wave-uniform: the records are addressed from scalar registers - 75 VGPRs before) */
#endif

QZ_KERNEL_OCC(64 * QZK_RES_WAVES, QZK_RES_OCC) qzk_lz_resolve_kernel(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs,
                                const qzk_tokseg *ts, uint32_t ts_stride, const uint8_t *lits, const qzk_seq *seqs,
                                const qzk_chain *chains, const uint32_t *order /* or NULL */, uint32_t count)
{
    /* `order`: the launch covers `count` segments picked by index (a range of the OUTPUT, so that it can leave for the
     * host while the next range is resolved); NULL: all nsegs in array order */
    QZ_LDS __attribute__((aligned(16))) uint8_t obuf_all[QZK_RES_WAVES][QZK_RB_OB];
    QZ_LDS __attribute__((aligned(16))) uint8_t lbuf_all[QZK_RES_WAVES][QZK_RB_LT];
#ifdef QZK_SPEC_PROF
    if ((threadIdx.x & 63) == 0) atomicMin(&qzk_stamp_b[0], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    const int lane = qz_lane();
    const uint32_t widx = blockIdx.x * QZK_RES_WAVES + (threadIdx.x >> 6);
    if (widx >= (order ? count : nsegs)) return;
    const uint32_t sidx = qz_uniform(order ? order[widx] : widx);  /* wave-uniform: the segment's records are addressed from scalar registers */
    if (sidx >= nsegs) return;
    const qzk_infseg sg = segs[sidx];
    if (res[sidx].status < 0 || (sg.flags & QZK_INF_COUNT_ONLY)) return;
    qzk_rb S;
    qzk_rb_init(&S, obuf_all[threadIdx.x >> 6], lbuf_all[threadIdx.x >> 6], out + sg.out_off,
                (sg.flags & QZK_INF_THROUGH_FLUSH) ? sg.out_off : 0 /* bytes a match may reach back before the segment */, sg.out_cap);
    const uint32_t nel = chains[sidx].nel;
    int err = 0;
    for (uint32_t e = 0; e < nel && !err; e++) {
        const qzk_chain_el ce = chains[sidx].el[e];
        if (ce.sub == QZK_PIECE_RAW) {                  /* a stored block: copy it from the compressed input */
            if ((uint64_t)S.obase + ce.seq_count > sg.out_cap) { err = QZK_INF_EOUT; break; }
            qzk_rb_flush(&S, lane);
            qzk_rb_direct(&S, comp + sg.in_off + ce.seq_first, ce.seq_count, lane);
            QZK_RSYNC();
            qzk_rb_skip(&S, ce.seq_count);
            continue;
        }
        const qzk_tokseg tk = ts[(uint64_t)sidx * ts_stride + ce.sub];
        const uint8_t *lp = lits + tk.lit_off + ce.lit_first;
        const qzk_seq *sq = seqs + tk.seq_off - 1 - ce.seq_first;       /* the piece's first record; the next ones BELOW it */
        const uint32_t ns = ce.seq_count;
        uint32_t lbase = 0;                             /* literal bytes of this piece consumed by earlier batches */
        uint32_t b0 = 0;
        uint64_t cur = (uint32_t)lane < ns ? qzk_rb_ld64((const uint8_t *)(sq - lane)) : 0;
        while (b0 < ns) {
            /* a record: {u32 literal run, u16 match length, u16 distance - 1} */
            uint32_t litrun = (uint32_t)cur, mlen = (uint32_t)(cur >> 32) & 0xffffu, dist = (uint32_t)(cur >> 48) + 1;
            if (b0 + (uint32_t)lane == 0) litrun -= ce.lrun_skip;      /* literals the sub-decoder produced before it was in step */
            uint32_t s_tot = qzk_wave_scan_incl(litrun + mlen, lane), s_lit = qzk_wave_scan_incl(litrun, lane);
            uint32_t n = qzk_rb_fit(s_tot, s_lit);
            if (n > ns - b0) n = ns - b0;
            if (n == 0) {
                /* the first sequence alone is more than a batch holds (a long literal run): straight to the output */
                const uint32_t L = qz_readlane(litrun, 0), M = qz_readlane(mlen, 0), D = qz_readlane(dist, 0);
                if ((uint64_t)S.obase + L + M > sg.out_cap) { err = QZK_INF_EOUT; break; }
                if (M != 0 && (uint64_t)D > (uint64_t)S.obase + L + S.hist) { err = QZK_INF_EHIST; break; }
                qzk_rb_flush(&S, lane);
                qzk_rb_direct(&S, lp + lbase, L, lane);
                QZK_RSYNC();
                qzk_rb_skip(&S, L);
                if (M) { qzk_rb_direct_match(&S, D, M, lane); QZK_RSYNC(); qzk_rb_skip(&S, M); }
                lbase += L; b0 += 1;
                cur = b0 + (uint32_t)lane < ns ? qzk_rb_ld64((const uint8_t *)(sq - b0 - lane)) : 0;
                continue;
            }
            /* the records of the batch after this one: asked for now, used when this batch is out */
            const uint64_t nxt = b0 + n + (uint32_t)lane < ns ? qzk_rb_ld64((const uint8_t *)(sq - b0 - n - lane)) : 0;
            if ((uint32_t)lane >= n) { s_tot -= litrun + mlen; s_lit -= litrun; litrun = 0; mlen = 0; }    /* (sums of idle lanes: never read) */
            const uint32_t Tb = qz_readlane(s_tot, (int)n - 1), Lb = qz_readlane(s_lit, (int)n - 1);
            err = qzk_rb_batch(&S, lp + lbase, Lb, s_lit - litrun, litrun, mlen, dist, s_tot, Tb, lane);
            if (err) break;                             /* (without this the next batch's 0 took its place: a match that reached
                                                         * before the segment, decoded by a lane other than lane 0, went through
                                                         * as a success - found by tools/sim_fuzz_corrupt.py, seed 188949) */
            lbase += Lb; b0 += n;
            cur = nxt;
        }
    }
    if (!err) qzk_rb_flush(&S, lane);
    if (err) res[sidx].status = err;                    /* wave-uniform value, every lane stores the same word */
}

#endif
