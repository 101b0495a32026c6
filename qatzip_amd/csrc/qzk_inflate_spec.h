/*
 * qzk_inflate_spec.h — K3b phase A with K LANES PER SEGMENT, gfx950.
 *
 * Phase A of the two-phase inflate (qzk_inflate_lane.h) is bound twice over: the LDS holds the root tables of 128
 * segments per CU, so a wave carries sixteen decoding lanes out of sixty-four - and every wave instruction costs the
 * same four cycles whether sixteen lanes follow it or all of them; and a lane walks its segment alone, so the phase
 * cannot end before the longest segment's chain does (a 1 GiB call of 64 KB segments takes as long as a 2 GiB one).
 * Huffman streams resynchronise on their own: a decoder started at an arbitrary bit falls into step with the true
 * symbol boundaries after a few symbols (codes of very even lengths take hundreds; only a perfectly flat code never
 * does).  So a segment is decoded by a GROUP of K lanes that share its tables - the wave's other lanes, which the LDS
 * left idle - block by block ("rounds"):
 *   1. lane 0 of the group parses the block header where the previous round ended (stored blocks become raw pieces:
 *      they are a memcpy for phase B) and builds the Huffman tables in the group's LDS;
 *   2. the K lanes start at K evenly spaced bit offsets of what is left of the segment and decode with the serial
 *      kernel's own trip (qzk_lane_trip) into sub-streams of their own.  Every lane leaves a trail: the bit position
 *      and token counters of its trip starts - the first QZK_SPEC_DENSE one by one, then every QZK_SPEC_EVERY-th, as
 *      long as it runs.  A lane that has reached a neighbour's share looks at that trail: when the next mark is within
 *      reach of a trip it takes one symbol a trip (every symbol boundary is then a trip start), and the moment its own
 *      trip start IS a mark the two are in step for good - same tables, same bits - so the lane stops and the
 *      neighbour's tokens take over from that mark.  Lane 0 is right by construction, hence by induction every lane on
 *      the chain is.  A lane that is not in step by the end of a neighbour's share simply carries on into the next
 *      one's (round 3's version gave up there and handed the segment back: the segments that never fell into step
 *      were decoded twice, and set the pace).  Round 5 bounds what a lane can waste: one whose share begins behind
 *      the block's END_BLOCK - known as soon as a lane ON THE TRUE STREAM has met it - leaves at its next marked trip,
 *      and no lane goes more than QZK_SPEC_REACH shares without falling into step (what is left is shared out again);
 *   3. lane 0 walks the chain (the stops travel through cross-lane reads, not memory), appends the pieces to the
 *      segment's piece list (what phase B, qzk_lz_resolve_kernel, stitches together) and takes over the bit position
 *      behind the block's END_BLOCK for the next round.  A lane whose piece ended up on nobody's chain takes it back
 *      (its sub-stream's counters return to where the round began): junk no longer fills the scratch of a lane that the
 *      segment's later blocks need.
 * What does not fit (bad data, a sub-stream outgrowing its scratch, too many pieces) is answered with QZK_INF_ESPEC
 * and the host decodes that segment with the serial kernel: this kernel delivers a validated segment or nothing.
 * Same place in the reference as the rest of K3: zlib inflate(), src/qatzip_sw.c:339.
 */
#ifndef QZK_INFLATE_SPEC_H
#define QZK_INFLATE_SPEC_H
#include "qzk_inflate_lane.h"

#ifndef QZK_SPEC_NREC
#define QZK_SPEC_NREC 64           /* marks a lane may leave per round ... */
#endif
#ifndef QZK_SPEC_DENSE
#define QZK_SPEC_DENSE 24          /* ... its first trips one by one ... */
#endif
#ifndef QZK_SPEC_EVERY
#define QZK_SPEC_EVERY 32          /* ... then one trip in so many (a power of two) */
#endif
#ifndef QZK_SPEC_REACH
#define QZK_SPEC_REACH(K) ((K) >= 16 ? 6u : 4u)    /* shares a lane decodes from its start before it gives the rest back */
#endif
#define QZK_SPEC_MINBITS 512u      /* a block is split only when every lane gets at least this much of it */
/* scratch of the sub-streams, sized by NEED (round 6; rounds 4-5 sized every one for its worst case - lane 0 for a whole
 * segment, the others for two or K - 1 shares of "a match every three bytes", literals and sequences apart: 673 KB per
 * 64 KB segment at K = 4, 44 GB for a 4 GiB call).  A lane's sub-stream is one region (qzk_inflate_lane.h: literals up,
 * sequences down) of QZK_SPEC_SLACK8 eighths of a share (a lane decodes past its own share until it falls into step, the
 * last one to the block's end) at QZK_SPEC_DENS8 eighths of a byte of scratch per byte of output - zlib's level-1 parse of
 * the bench's text segments needs 1.4 (a fifth of the bytes literals, a match every 6.7), its random segments 1.0.  What
 * outgrows that is not lost: a lane out of scratch stops, is on nobody's chain, and the block goes on in a round of its
 * own; a segment whose lane 0 runs out is handed back (QZK_INF_ESPEC) and the serial kernel decodes it into a HAND-BACK
 * area of whole-segment regions the host keeps for one segment in thirty-two (more than that: the call takes the one-lane
 * path, whose single region per segment holds any segment and is 3.7 times the output). */
#ifndef QZK_SPEC_DENS8
#define QZK_SPEC_DENS8 15
#endif
#ifdef QZK_SPEC_TINY        /* emulator stress builds: sub-streams that overflow at once, so that the continuation rounds are exercised */
#define QZK_SPEC_SLACK8(K, j) (8ull / 3)
#else
/* eighths of a share: 1.25 shares, the last lane 1.75; lane 0 two and a half and never less than half of the segment - it is the lane
 * that is right by construction and decodes on through the share of a neighbour that met a false END_BLOCK after one trip.  Sixteen
 * lanes (segments of 16 - 128 KB in launches that do not fill the chip) get one and a half, thirty-two (256 / 512 KB segments: a score
 * of blocks each, lanes that go up to six small shares before they fall into step) two: 1 GiB of 128 KB segments 11.5 -> 10.7 ms of
 * phase A, of 512 KB 55 -> 18.  All lanes together: 1.66 - 1.92 segments' worth up to K = 16, 2.5 at K = 32 */
#define QZK_SPEC_SLACK8(K, j) ((j) == 0 ? ((K) > 5 ? 4ull * (K) : 20ull) : (K) >= 32 ? ((j) == (K) - 1 ? 20ull : 16ull) : (j) == (K) - 1 ? 14ull : (K) >= 16 ? 12ull : 10ull)
#endif
/* a sub-stream's region in bytes (a multiple of 64): the margins are what QZK_SPEC_EVERY unchecked trips can add, both ways */
#define QZK_SPEC_MARGIN (96 + 512 + 8 * (10 + 64))
#define QZK_SPEC_REGION(out_cap, K, j) ((((uint64_t)(out_cap) * QZK_SPEC_DENS8 / 8 * QZK_SPEC_SLACK8(K, j) / 8 / (K) + 63) & ~(uint64_t)63) + ((QZK_SPEC_MARGIN + 63) & ~63))
#define QZK_SPEC_HANDBACK(nsegs) ((nsegs) / 32u > 64u ? (nsegs) / 32u : 64u)      /* whole-segment regions kept for the segments handed back */

typedef struct __attribute__((aligned(32))) { uint32_t pos, nlit, nseq, lrun, olen, tag, pad0, pad1; } qzk_rec;     /* a mark: trip start (bit offset) and the token counters there */
enum { QZK_ST_RUN = 0, QZK_ST_SYNC, QZK_ST_EOB, QZK_ST_REDO };
/* how a lane's round ended: SYNC with mark cidx of lane `target`, or EOB (its block ended at bit `at`), or REDO (why) */
typedef struct { uint32_t kind, target, cidx, at, nlit0, nseq0, olen0, nlit, nseq, olen; } qzk_spec_stop;

#ifdef QZ_SIM
QZ_DEV uint32_t qzk_ld32_l2(const uint32_t *p) { return *p; }
#else
/* served by the L2: marks are written by another lane of the wave, possibly a moment ago */
QZ_DEV uint32_t qzk_ld32_l2(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

#ifdef QZK_SPEC_STATS
static uint32_t qzk_spec_stats[1 << 20];
#endif
#ifdef QZK_SPEC_PROF           /* profiling builds only (tools/prof_spec.py): shader clocks of the hot loop's parts, per wave */
__device__ unsigned long long qzk_spec_prof[8192][8];
__device__ unsigned int qzk_spec_seg[1 << 17];      /* per segment (launch place): shader clocks / 64 from its wave's start to the segment's end */
__device__ unsigned long long qzk_stamp[8];     /* wall-clock (100 MHz) first entry / last exit of: marker scan, phase A, phase B */
#define QZK_STAMP_IN(k) do { if ((threadIdx.x & 63) == 0) atomicMin(&qzk_stamp[2 * (k)], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define QZK_STAMP_OUT(k) do { if ((threadIdx.x & 63) == 0) atomicMax(&qzk_stamp[2 * (k) + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define QZK_PCLK() ((unsigned long long)__builtin_readcyclecounter())
#define QZK_SPROF(k) do { const unsigned long long c_ = QZK_PCLK(); pacc[k] += c_ - pt; pt = c_; } while (0)
/* the looks at a neighbour's trail: a trip in which ANY lane of the wave looks costs the whole wave the look (two loads from
 * L2 and their wait); plook = the wave's clocks in there, nlook = such trips */
#define QZK_LOOK_IN() const bool any_ = qz_ballot(at_ >= wake) != 0; const unsigned long long lk_ = any_ ? QZK_PCLK() : 0ull
#define QZK_LOOK_OUT() do { if (any_) { plook += QZK_PCLK() - lk_; nlook++; } } while (0)
#else
#define QZK_SPROF(k) ((void)0)
#define QZK_LOOK_IN() ((void)0)
#define QZK_LOOK_OUT() ((void)0)
#endif
/* OCC = waves per SIMD the register budget is cut for.  Two: ~237 VGPRs, nothing spilled - what the kernel wants.  Three: 168
 * VGPRs, ~95 spilled (230 bytes of scratch a lane) - and still the better deal for a launch of more than two rounds of waves,
 * because the trip is a dependent chain and a third wave fills what two leave idle (round 6, profiles/r6_phaseA_order.txt:
 * 4 GiB of 64 KB segments, eight lanes each, 18.7 -> 15.6 ms; four waves a SIMD spill 290 and lose).  K = 4 cannot: its
 * sixteen rows of root tables are an eighth of the CU's LDS. */
template <int K, int OCC>
QZ_KERNEL_OCC(64, OCC) qzk_inflate_spec_kernel(const uint8_t *comp, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs,
                                  qzk_inf_tab *tabs, const qzk_tokseg *ts /* [nsegs * K] */, uint8_t *lits, qzk_seq *seqs,
                                  qzk_chain *chains, qzk_rec *recs /* [nsegs * K * QZK_SPEC_NREC] */, uint64_t epoch /* of the launch, process-wide, never reused */,
                                  uint32_t over_shares /* shares beyond its own the last lane of a block may decode before the rest is shared out again */)
{
    constexpr int SPW = 64 / K;                                     /* segments per wave */
    QZ_LDS uint16_t roots[SPW][QZK_LANE_ROOTSZ];
    /* Round 5: where the group's block ENDS, as soon as a lane has seen its END_BLOCK (the lowest such position: a lane that
     * decodes garbage can only claim an end beyond its own start).  The shares of a round are a guess of the block's length,
     * and a lane whose share begins beyond the real end decodes the NEXT blocks' bits with this block's tables: nothing ever
     * puts it in step, it ran to the end of the segment's input - 1400 to 5800 trips where its group-mates made 270 (the
     * emulator's trip counts over the bench data, profiles/r5_phaseA_tail.txt), the whole wave waiting, and the junk tokens
     * filled its scratch, so that it sat out the segment's later blocks.  Now such a lane leaves at its next marked trip. */
    /* (K = 4: these live in the spare bytes behind each segment's root tables, qzk_inflate_lane.h - they are written after a round's
     * headers and dead before the next ones, when the code-length root takes the whole side region) */
    constexpr bool OVL = 8 + 5 * K <= QZK_SIDE_SPARE;
    QZ_LDS uint32_t gend_s[OVL ? 1 : SPW];
    /* Only a lane that decodes the TRUE stream may say where the block ends (a lane not yet in step meets a false END_BLOCK
     * once in ~30 000 symbols: some forty times per 64 MiB).  Who is on the true stream is known link by link: lane 0 is; a lane
     * that fell into step with lane t's trail vouches for t from there on, if it is vouched for itself.  Every lane leaves its
     * link (gnext) and its END_BLOCK (geob) when it stops; whoever is confirmed walks the links in front of it. */
    QZ_LDS uint32_t gconf_s[OVL ? 1 : SPW], geob_s[OVL ? 1 : 64];
    QZ_LDS uint8_t gnext_s[OVL ? 4 : 64];
#ifdef QZK_SPEC_PROF
    QZK_STAMP_IN(1);
#endif

    const int lane = (int)threadIdx.x, g = lane / K, j = lane % K, gbase = lane - j;
    /* (the host lists the segments longest compressed length first, qzd_inflate.hip inflate_stream: the segments of a wave
     * are of one size and the launch does not end with the long ones) */
    const uint32_t sidx = blockIdx.x * SPW + (uint32_t)g;
    const bool live = sidx < nsegs;
    const qzk_infseg sg = segs[live ? sidx : 0];
    qzk_inf_tab *T = tabs + (live ? sidx : 0);
    uint16_t *const lroot = roots[g], *const droot = lroot + (1 << QZK_LLROOT);
    /* my group's: where the block ends, who is confirmed, the lanes' END_BLOCKs [K] and links [K] */
    uint8_t *const spare = (uint8_t *)droot + QZK_SIDE_BYTES - QZK_SIDE_SPARE;
    uint32_t *const gend = OVL ? (uint32_t *)spare : &gend_s[OVL ? 0 : g], *const gconf = OVL ? (uint32_t *)(spare + 4) : &gconf_s[OVL ? 0 : g];
    uint32_t *const geob = OVL ? (uint32_t *)(spare + 8) : &geob_s[OVL ? 0 : gbase];
    uint8_t *const gnext = OVL ? spare + 8 + 4 * K : &gnext_s[OVL ? 0 : gbase];
    const uint32_t slot = (live ? sidx : 0) * K + (uint32_t)j;     /* my sub-stream */
    qzk_rec *const myrec = recs + (uint64_t)slot * QZK_SPEC_NREC;
    qzk_chain *C = chains + (live ? sidx : 0);

    qzk_lane_st S;
    S.b.p = comp + sg.in_off; S.b.end = sg.in_len; qzk_lseek(&S.b, 0);
    S.op = 0; S.nblocks = 0; S.last = 0; S.clen = 0; S.rpos = 0;
    S.out_cap = j == 0 ? sg.out_cap : 0xffffffffu;                 /* lane 0 knows the output offset, phase B checks the rest */
    S.lmax = 0; S.dmax = 0; S.lbase = 0; S.status = QZK_INF_EDATA; S.state = QZK_LS_HDR;
    S.through = false;
    qzk_tok_out O;
    qzk_tok_init(&O, lits + ts[slot].lit_off, seqs + ts[slot].seq_off, false);
    /* checked on the trips that leave a mark: the margin is what QZK_SPEC_EVERY trips can add */
    /* literal bytes + 8 x sequences my region holds, less what the unchecked trips between two marks can add */
    const uint32_t tok_cap = (uint32_t)QZK_SPEC_REGION(sg.out_cap, K, j) - QZK_SPEC_MARGIN;
#define QZK_TOK_FULL(O_) (QZK_NLIT(O_) + 8u * (O_).nseq > tok_cap)
    /* where the segment's input is taken to end: the host's hint (the next candidate's start), moved on by lane 0 when the
     * decode gets there and the segment goes on - 00 00 FF FF inside a segment's own data cuts its hint short, and a lane 0
     * left to decode the rest alone was a whole serial chain at the end of the launch (profiles/r4_phaseA_timeline.txt) */
    const uint32_t end_bits = sg.in_len > 0x1fffffffu ? 0xffffffffu : 8u * sg.in_len;
    uint32_t limit_bits = 8u * (sg.pad != 0 && sg.pad < sg.in_len ? sg.pad : sg.in_len);
    uint32_t LR[QZK_LR_WORDS], DR[QZK_DR_WORDS];
    for (int i = 0; i < QZK_LR_WORDS; i++) LR[i] = 0;
    for (int i = 0; i < QZK_DR_WORDS; i++) DR[i] = 0;
    qzk_dsyms DS; DS.w0 = DS.w1 = DS.w2 = 0;

    /* lane 0 only: the segment's result so far */
    uint32_t nel = 0, total_out = 0, blk_bits = 0;                  /* pieces, output bytes, length of the previous Huffman block */
    int seg_status = QZK_INF_ESPEC;                                 /* set to FINAL / FLUSH when the segment ends well */
    uint32_t why = 0;                                               /* developer aid: why the segment was handed back */
    bool seg_done = !live;                                          /* lane 0: nothing more to do for this segment */
    uint32_t cont_at = 0, prev_span = 0, blk_at = 0; bool cont = false, cont_past = false, cont_grow = false;                        /* lane 0: the block goes on at this bit (a lane on the chain ran out of scratch) */
    uint32_t c_last = 0, c_lmax = 0, c_dmax = 0, c_lbase = 0;       /* ... with the tables it has */

#ifdef QZK_SPEC_PROF
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = QZK_PCLK(), plook = 0, nlook = 0;
    const unsigned long long p_start = pt; unsigned long long p_segdone = 0;
    const unsigned long long p_begin = __builtin_amdgcn_s_memrealtime();       /* the 100 MHz clock every wave agrees on */
#endif
    for (uint32_t round = 0;; round++) {
        /* ---- 1. lane 0: block headers and stored blocks up to the next Huffman block ---- */
        uint32_t h_mode = 0, h_end = 0, h_span = 0, h_last = 0, h_lmax = 0, h_dmax = 0, h_lbase = 0;
        if (j == 0 && cont) {
            /* the same block, from where the chain broke: no header, the tables stand; what is left is shared out again
             * (the lane whose scratch is full stops at its first trip and is on nobody's chain) */
            cont = false;
            h_mode = 2; h_end = cont_at; h_last = c_last; h_lmax = c_lmax; h_dmax = c_dmax; h_lbase = c_lbase;
            if (cont_past) {                                        /* a lane on the chain ran past the hint: as much again as the segment has taken so far */
                const uint32_t more = cont_at > 131072u ? cont_at : 131072u;      /* (16 KB at least: every round adds pieces) */
                limit_bits = end_bits - cont_at > more ? cont_at + more : end_bits;
            }
            uint32_t share = limit_bits > cont_at ? limit_bits - cont_at : 0;
            /* the block was longer than guessed (its last lane went a share beyond its own and was stopped there): what is
             * shared out now is twice as much, not all the rest - the guess doubles until it covers the block */
            if (cont_grow && prev_span != 0 && 2u * K * prev_span < share) share = 2u * K * prev_span;
            h_span = share > QZK_SPEC_MINBITS * K ? share / K : 0;
        } else if (j == 0) {
            while (!seg_done && h_mode == 0) {
                S.op = total_out;
                qzk_lane_header(&S, T, lroot, droot);
                if (S.state == QZK_LS_RAW) {
                    /* stored block: nothing to decode - phase B copies it straight from the input */
                    if (nel >= QZK_CHAIN_MAXEL) { why = 20; seg_done = true; break; }
                    qzk_chain_el e; e.sub = QZK_PIECE_RAW; e.seq_first = S.rpos; e.seq_count = S.clen; e.lit_first = 0; e.lrun_skip = 0;
                    C->el[nel++] = e;
                    total_out += S.clen; S.clen = 0;
                    if (S.last) { seg_status = QZK_INF_FINAL; seg_done = true; }
                    else S.state = QZK_LS_HDR;
                } else if (S.state == QZK_LS_SYM) {
                    const uint32_t at = 8u * S.b.pos - (uint32_t)S.b.bc;
                    h_mode = 1; h_end = at; blk_at = at; h_last = S.last; h_lmax = (uint32_t)S.lmax; h_dmax = (uint32_t)S.dmax; h_lbase = S.lbase;
                    /* what the group shares: the rest of the segment, or - from the second block on - a block as long as
                     * the previous one (zlib closes a block every 32767 symbols, so blocks of a segment are alike).
                     * Too little to share: span 0 parks the other lanes, lane 0 decodes alone */
                    uint32_t share = limit_bits > at ? limit_bits - at : 0;
                    /* more than 5 bits per output byte: mostly literals, the segment will take several 32767-symbol
                     * blocks - guess half of it for the first one */
#ifndef QZK_SPEC_WHOLE
                    if (!blk_bits && (uint64_t)limit_bits > 5ull * sg.out_cap) blk_bits = share / 2;
#endif
                    if (blk_bits && blk_bits + blk_bits / 8 < share) share = blk_bits + blk_bits / 8;
                    h_span = share > QZK_SPEC_MINBITS * K ? share / K : 0;
#ifdef QZK_X_ALONE
                    h_span = 0;
#endif
                } else if (S.state == QZK_LS_HDR) {
                    /* an empty stored block that is not the flush marker: next header */
                } else {                                            /* DONE: FINAL / FLUSH, or an error - lane 0 reads the true stream, so it is the segment's */
                    seg_status = S.status;
                    seg_done = true;
                }
            }
        }
        QZK_SPROF(0);                                               /* [0] headers */
#ifdef QZK_SPEC_PROF
        if (j == 0 && live && seg_done && !p_segdone) p_segdone = QZK_PCLK() - p_start;
#endif
        if (j == 0 && h_span != 0) prev_span = h_span;
        if (j == 0) { *gend = 0xffffffffu; *gconf = 1u; }
        gnext[j] = 0xff; geob[j] = 0xffffffffu;
        qz_wave_sync();                                             /* the tables (LDS) and their ranges (the segment's record) are the group's now */
        h_mode = qz_shfl(h_mode, gbase); h_end = qz_shfl(h_end, gbase); h_span = qz_shfl(h_span, gbase); h_last = qz_shfl(h_last, gbase);
        h_lmax = qz_shfl(h_lmax, gbase); h_dmax = qz_shfl(h_dmax, gbase); h_lbase = qz_shfl(h_lbase, gbase);
        limit_bits = qz_shfl(limit_bits, gbase);
        if (qz_ballot(h_mode != 0) == 0) break;                     /* every group of the wave has finished */

        /* ---- 2. the group decodes the block ---- */
        bool active = h_mode != 0 && (j == 0 || h_span != 0);
        const uint32_t my_at = h_end + (uint32_t)j * h_span;       /* where my share of this round begins */
        if (active && (j > 0 || h_mode == 2)) {                     /* my guessed start (lane 0 stands behind the header, or goes to where the chain broke) */
            const uint32_t at = h_end + (uint32_t)j * h_span;
            if (j > 0 && (at + 64 >= limit_bits || (at >> 3) + 64 > S.b.end || QZK_TOK_FULL(O))) active = false;
            else {
                qzk_lseek(&S.b, at >> 3);
                qzk_lrefill(&S.b);
                QZK_DROP(&S.b, at & 7);
                S.last = h_last; S.op = j == 0 ? total_out : 0;
            }
        }
        S.lmax = (int)h_lmax; S.dmax = (int)h_dmax; S.lbase = h_lbase;
        if (h_mode != 0) { qzk_longtab_load(LR, DR, T, S.lmax, S.dmax); qzk_dsyms_load(&DS, T->dsorted, T->dcount); }
        S.state = QZK_LS_SYM;
        QZK_SPROF(1);                                               /* [1] starts and tables */
        static_assert(QZK_CHAIN_MAXEL < 1024, "a round's tag has ten bits");
        /* a segment takes at most QZK_CHAIN_MAXEL rounds (each adds a piece): no two of a launch share a tag; the launch's
         * number is 54 bits wide across tag and pad0 - the 22 bits the tag alone has room for come round again after 4.2 million launches
         * (every lone 64 KB call is one), and a record left by the launch 2^22 before this one must not pass for a mark of this one */
        const uint32_t tag = ((uint32_t)epoch << 10) | (round & 1023u), tag_hi = (uint32_t)(epoch >> 22);
        /* a lane that started beyond the end of the block (the block was shorter than guessed) or that never falls into
         * step decodes garbage: the end of the segment's input stops it; lane 0 is always right */
        const uint32_t give_up = j == 0 ? 0xffffffffu : limit_bits + 64;
        /* the last lane decodes to the end of the block, however much longer than guessed the block is - but not alone for
         * long: a share beyond its own it stops, and what is left of the block is shared out again (a round of its own, the
         * guess doubled).  Blocks of one segment differ: 32767 symbols are a few KB of matches or 36 KB of literals. */
        uint32_t over = h_span != 0 && j == K - 1 && over_shares < 1000u ? h_end + ((uint32_t)K + over_shares) * h_span : 0xffffffffu;
        /* ... and no lane goes more than QZK_SPEC_REACH shares without falling into step with anybody: where the codes are
         * all of one length (bytes that do not compress) a decoder that started off a boundary stays off it for thousands of
         * symbols, and the shares of a short guess are far smaller than that distance - the lanes used to run until their
         * scratch was full, five times the trips of the round's useful ones (profiles/r5_phaseA_tail.txt).  Lane 0 too: what
         * it would decode alone is shared out again with the guess doubled.  (Measured, tools/run/r5_p.sh and r5_q.sh: 3 shares
         * break chains that 4 complete - 64 MiB: 7.2 against 3.7 ms; 8 for all lanes, or for the lanes known to be on the true
         * stream only, costs 1 GiB of 64 KB segments - eight lanes each - 10.4-11.1 ms against 7.9; with sixteen lanes a share is
         * half as long and 6 does better than 4: 1 GiB of 128 KB segments 10.6 against 11.8 ms.) */
        if (h_span != 0 && (uint64_t)my_at + (uint64_t)QZK_SPEC_REACH(K) * h_span < over) over = my_at + QZK_SPEC_REACH(K) * h_span;
        /* lane 0 knows where in the segment's output it stands, so a match that reaches back before the segment stops it at
         * once (the others are checked by phase B).  It is what ends the decode of a candidate that is no segment - 00 00 FF FF
         * inside compressed data: without it lane 0, which nothing else bounds, read garbage until 64 KB of output had come
         * together, a whole serial chain begun when the launch was nearly over (profiles/r4_phaseA_timeline.txt: 8 of 23 ms) */
        const uint64_t hist = j == 0 ? 0ull : (uint64_t)1 << 40;
        /* whose trail I am looking at: `target` (none before my own share ends), its mark `cursor`, that mark's position
         * `rp` (0xffffffff: none within sight) */
        uint32_t target = (uint32_t)j, cursor = 0, rp = 0xffffffffu;
        uint32_t wake = h_span != 0 && j + 1 < K ? h_end + ((uint32_t)j + 1) * h_span : 0xffffffffu;    /* the position at which I look at a trail next */
        uint32_t ntrip = 0, ridx = 0;
        qzk_spec_stop st; st.kind = QZK_ST_RUN; st.target = 0; st.cidx = 0; st.at = 0;
        st.nlit0 = QZK_NLIT(O) - O.lrun; st.nseq0 = O.nseq; st.olen0 = S.op;   /* where my piece of this round starts */
        st.nlit = st.nseq = st.olen = 0;

        /* What every trip start does before the trip.  Every trip: am I on the neighbour's mark (in step: stop), has a mark or
         * the next lane's share come up (look at the trail), may the trip take literals behind its first symbol (`allow`: not
         * within reach of a mark, where every symbol boundary must be a trip start).  On the trips that leave a mark - the
         * lanes of a wave count their trips together, so this is a branch the wave takes as one - the mark is stored, the
         * scratch and the end of the input are checked (with the margin of the trips in between) and a trail that was not
         * written yet is looked at again.  The mark goes out BEFORE the trip's own two unconditional stores: the wait at the
         * top of the loop counts those two, and a third store in front of them only makes it wait for an older one. */
#define QZK_SPEC_PRE() \
        const uint32_t at_ = 8u * S.b.pos - (uint32_t)S.b.bc; \
        const bool due_ = ntrip < QZK_SPEC_DENSE || (ntrip & (QZK_SPEC_EVERY - 1)) == 0;    /* (the lanes of a wave count together) */ \
        ntrip++; \
        if (due_) { \
            if (ridx + 1 < QZK_SPEC_NREC) { \
                qzk_rec r_; r_.pos = at_; r_.nlit = QZK_NLIT(O); r_.nseq = O.nseq; r_.lrun = O.lrun; r_.olen = S.op; r_.tag = tag; r_.pad0 = tag_hi; r_.pad1 = 0; \
                myrec[ridx++] = r_; \
            } \
            if (st.kind == QZK_ST_RUN && (QZK_TOK_FULL(O) || at_ >= give_up || at_ >= over)) { \
                st.kind = QZK_ST_REDO; st.cidx = at_ >= give_up ? 1u : at_ >= over ? 4u : 2u; st.at = at_; } \
            if (st.kind == QZK_ST_RUN && j != 0 && my_at >= *(volatile uint32_t *)gend) {      /* my share lies behind the block's end: I am decoding garbage */ \
                st.kind = QZK_ST_REDO; st.cidx = 5u; st.at = at_; } \
            if (rp == 0xffffffffu && target != (uint32_t)j) wake = 0;           /* the neighbour may have written since */ \
        } \
        QZK_LOOK_IN(); \
        if (at_ >= wake) { \
            while (h_span != 0 && target + 1 < (uint32_t)K && at_ >= h_end + (target + 1) * h_span) { target++; cursor = 0; } \
            const uint32_t next_terr_ = h_span != 0 && target + 1 < (uint32_t)K ? h_end + (target + 1) * h_span : 0xffffffffu; \
            rp = 0xffffffffu; \
            bool again_ = false;                                    /* more marks to step over than one look takes: look again at the next trip */ \
            if (target != (uint32_t)j) { \
                again_ = true; \
                for (int tries_ = 0; tries_ < 4; tries_++) { \
                    const qzk_rec *q_ = recs + ((uint64_t)(sidx * K + target) * QZK_SPEC_NREC + cursor); \
                    const uint32_t qp_ = qzk_ld32_l2(&q_->pos), qt_ = qzk_ld32_l2(&q_->tag); \
                    if (qt_ != tag || qzk_ld32_l2(&q_->pad0) != tag_hi) { again_ = false; break; }     /* not written (yet): the neighbour has not got there, or has stopped */ \
                    if (qp_ >= at_) { rp = qp_; again_ = false; break; } \
                    if (cursor + 2 >= QZK_SPEC_NREC) { again_ = false; break; }    /* the trail ends here */ \
                    cursor++; \
                } \
            } \
            wake = again_ ? 0u : rp != 0xffffffffu && rp + 1 < next_terr_ ? rp + 1 : next_terr_; \
            QZK_PIN(rp); QZK_PIN(wake); QZK_PIN(cursor);            /* the waits for the marks just read belong in here */ \
        } \
        QZK_LOOK_OUT(); \
        if (st.kind == QZK_ST_RUN && at_ == rp) { st.kind = QZK_ST_SYNC; st.target = target; st.cidx = cursor; } \
        const bool allow = rp - at_ > 56u;                          /* a mark within reach: one symbol a trip, so that no boundary is stepped over */

        while (qz_ballot(active) != 0) {
            if (active) {
                qzk_lbits *b = &S.b;
                if (b->pos + 64 <= b->end) {
                    b->pos -= (uint32_t)(b->bc >> 3); b->bc &= 7; b->bb &= (1ull << b->bc) - 1;
                    qzk_win W; qzk_win_init(&W, b->p, b->pos);
#ifdef QZK_SPEC_PROF
                    const unsigned long long pl0 = QZK_PCLK();
#endif
                    for (int trip = 0; trip < QZK_TOK_TRIPS && st.kind == QZK_ST_RUN && S.state == QZK_LS_SYM && b->pos + 64 <= b->end; trip++) {
                        const bool mv = qzk_win_step(&W, b->pos);   /* the lane's memory traffic first (qzk_inflate_lane.h): window, pieces, next load */
                        qzk_tok_drain(&O);
                        qzk_win_load(&W, b->p, mv);
                        QZK_SPEC_PRE();
                        if (st.kind == QZK_ST_RUN) {
                            b->bb |= qzk_win_get(&W, b->pos) << b->bc;
                            b->pos += (uint32_t)(63 - b->bc) >> 3; b->bc |= 56;
                                qzk_lane_trip<QZK_LIT_RUN - 1>(&S, &O, T, lroot, droot, hist, LR, DR, &DS, allow);
    #ifdef QZK_SPEC_PROF
                            pacc[7]++;
#endif
                        }
                    }
#ifdef QZK_SPEC_PROF
                    pacc[6] += QZK_PCLK() - pl0;                    /* this lane's clocks inside the hot loop */
#endif
                    b->pos -= (uint32_t)(b->bc >> 3); b->bc &= 7; b->bb &= (1ull << b->bc) - 1;
                    const uint32_t keep = (uint32_t)b->bc; const uint64_t low = b->bb;
                    qzk_lseek(b, b->pos);
                    b->bb = low; b->bc = (int)keep;
                } else {
                    for (int trip = 0; trip < 64 && st.kind == QZK_ST_RUN && S.state == QZK_LS_SYM; trip++) {
                        QZK_SPEC_PRE();
                        (void)allow;
                        if (st.kind == QZK_ST_RUN) {
                            qzk_lrefill(b);
                            qzk_lane_symbol<true, false>(&S, &O, T, lroot, droot, hist);
                        }
                    }
                }
                if (st.kind == QZK_ST_RUN && S.state != QZK_LS_SYM) {
                    /* END_BLOCK leaves HDR (or DONE with FINAL when the block was the last one); anything else is an error */
                    if (S.state == QZK_LS_HDR || (S.state == QZK_LS_DONE && S.status == QZK_INF_FINAL)) {
                        st.kind = QZK_ST_EOB; st.at = 8u * b->pos - (uint32_t)b->bc;
                    } else { st.kind = QZK_ST_REDO; st.cidx = 3u; st.at = (uint32_t)S.status; }
                }
                if (st.kind != QZK_ST_RUN) {
                    if (O.lrun) qzk_tok_seq(&O, 0u, 0u);            /* my piece ends with its pending literals */
                    qzk_tok_flush(&O);                              /* ... and is in memory whole: a later round's may be taken back */
                    st.nlit = QZK_NLIT(O); st.nseq = O.nseq; st.olen = S.op;
                    active = false;
                    /* my link and my END_BLOCK for whoever confirms me; if I am confirmed already, I confirm what lies in front
                     * of me (the lanes of a wave run in lockstep and this block has no wave-wide operation in it: nobody else's
                     * bookkeeping interleaves with mine) */
                    if (st.kind == QZK_ST_SYNC) gnext[j] = (uint8_t)st.target;
                    if (st.kind == QZK_ST_EOB) geob[j] = st.at;
                    if ((*(volatile uint32_t *)gconf >> j) & 1u) {
                        uint32_t t = (uint32_t)j;
                        for (int hop = 0; hop < K; hop++) {
                            const uint32_t e = *(volatile uint32_t *)&geob[t];
                            if (e != 0xffffffffu) atomicMin(gend, e);           /* the block's end, from a lane that is right */
                            const uint32_t nx = *(volatile uint8_t *)&gnext[t];
                            if (nx >= (uint32_t)K || nx == t) break;
                            atomicOr(gconf, 1u << nx);
                            t = nx;
                        }
                    }
                }
            }
        }
#undef QZK_SPEC_PRE
        QZK_SPROF(2);                                               /* [2] the decode phase of the round (to its slowest lane) */
#if defined(QZ_SIM) && defined(QZK_SPEC_STATS)
        if (getenv("QZDBG_SPEC") && live && sidx == (uint32_t)atoi(getenv("QZDBG_SPEC")))
            fprintf(stderr, "round %u lane %2d: start %8u span %6u stop kind %u cidx %u at %8u trips %5u  target %u cursor %u rp %u wake %u limit %u nlit %u/%u nseq %u/%u ridx %u\n", round, j, my_at, h_span, st.kind, st.cidx,
                    st.kind == QZK_ST_RUN ? 0u : st.at, ntrip, target, cursor, rp, wake, limit_bits, QZK_NLIT(O), tok_cap, O.nseq, tok_cap / 8, ridx);
#endif
#ifdef QZK_SPEC_STATS          /* emulator builds only (tools/spec_stats.py): trips of every lane and round */
        if (live && round < 8) qzk_spec_stats[(sidx * K + (uint32_t)j) * 8 + round] = (ntrip & 0xffffffu) | (st.kind == QZK_ST_REDO ? (st.cidx & 15u) << 24 : 0u) | st.kind << 28;
#endif
        qz_wave_sync();
        /* ---- 3. lane 0 walks the chain of this round; the stops of the lanes it visits come through cross-lane reads ---- */
        {
            uint32_t cur = 0, c_nlit = 0, c_nseq = 0, c_lrun = 0, c_olen = 0, onchain = 0;
            bool from_rec = false, ok = false, walking = j == 0 && h_mode != 0;
            int bad_status = 0;
            for (int hop = 0; hop < K; hop++) {
                const int src = gbase + (int)(cur < (uint32_t)K ? cur : 0u);
                qzk_spec_stop s;
                s.kind = qz_shfl(st.kind, src); s.target = qz_shfl(st.target, src); s.cidx = qz_shfl(st.cidx, src); s.at = qz_shfl(st.at, src);
                s.nlit0 = qz_shfl(st.nlit0, src); s.nseq0 = qz_shfl(st.nseq0, src); s.olen0 = qz_shfl(st.olen0, src);
                s.nlit = qz_shfl(st.nlit, src); s.nseq = qz_shfl(st.nseq, src); s.olen = qz_shfl(st.olen, src);
                if (!walking) continue;
                const bool broke = s.kind == QZK_ST_REDO && (s.cidx == 4u || ((s.cidx == 1u || s.cidx == 2u || s.cidx == 5u) && cur != 0));
                if (s.kind != QZK_ST_SYNC && s.kind != QZK_ST_EOB && !broke) {
                    /* a lane on the chain decodes the true stream: what stopped it (cidx 3: bad data, the end of the input, the
                     * capacity) is the segment's own error; lane 0 out of scratch (its sub-stream holds a whole segment) or
                     * anything else is this kernel's */
                    why = 10 + s.cidx; walking = false;
                    if (s.kind == QZK_ST_REDO && s.cidx == 3u) bad_status = (int)s.at;
                    continue;
                }
                if (nel >= QZK_CHAIN_MAXEL) { why = 20; walking = false; continue; }
                qzk_chain_el e; e.sub = cur;
                if (from_rec) { e.seq_first = c_nseq; e.lit_first = c_nlit; e.lrun_skip = c_lrun; total_out += s.olen - c_olen; }
                else { e.seq_first = s.nseq0; e.lit_first = s.nlit0; e.lrun_skip = 0; total_out += s.olen - s.olen0; }
                e.seq_count = s.nseq - e.seq_first;
                C->el[nel++] = e;
                onchain |= 1u << cur;
                if (broke) {
                    /* the lane's sub-stream is full (or it ran past the input it was told of): its piece stands, and the block
                     * goes on from where it stopped in a round of its own - lane 0's sub-stream has room for all of it */
                    ok = true; cont = true; cont_at = s.at; cont_past = s.cidx == 1u; cont_grow = s.cidx == 4u;
                    c_last = h_last; c_lmax = h_lmax; c_dmax = h_dmax; c_lbase = h_lbase;
                    walking = false;
                    continue;
                }
                if (s.kind == QZK_ST_EOB) {
                    ok = true; blk_bits = s.at - blk_at;            /* (the whole block, not its last round's part: a guess from that left lane 0 alone with the next one) */
                    qzk_lseek(&S.b, s.at >> 3); qzk_lrefill(&S.b); QZK_DROP(&S.b, s.at & 7);   /* I carry on behind the block */
                    if (h_last) { seg_status = QZK_INF_FINAL; seg_done = true; } else S.state = QZK_LS_HDR;
                    walking = false;
                    continue;
                }
                const qzk_rec *q = recs + ((uint64_t)(sidx * K + s.target) * QZK_SPEC_NREC + s.cidx);
                cur = s.target; from_rec = true;
                c_nlit = qzk_ld32_l2(&q->nlit); c_nseq = qzk_ld32_l2(&q->nseq); c_lrun = qzk_ld32_l2(&q->lrun); c_olen = qzk_ld32_l2(&q->olen);
            }
            /* a lane whose piece of this round is on nobody's chain takes it back: what it decoded was off the symbol
             * boundaries (or behind the block's end) and nobody will read it - left in place it filled the lane's sub-stream,
             * and a lane without scratch sits out every later round of the segment (lane 0 was seen decoding a whole block
             * alone behind fifteen full lanes).  The piece began at (nlit0, nseq0) with nothing pending; the 16 bytes and the
             * pair it began inside are in memory since the lane last stopped */
            onchain = qz_shfl(onchain, gbase);
            if (live && j != 0 && st.kind != QZK_ST_RUN && !((onchain >> j) & 1u)) {
                O.lw = st.nlit0; O.nseq = st.nseq0; O.lrun = 0; O.lfull = 0; O.qfull = 0;
                O.l0 = O.l1 = O.l2 = O.l3 = O.l4 = 0; O.q0 = O.q1 = 0;
                const uint32_t c = O.lw & 15u;
                if (c) {
                    const qzk_u32x4 v = *(const qzk_u32x4 *)(O.lp + (O.lw & ~15u));
                    const uint32_t w = c >> 2, m = (1u << (8u * (c & 3u))) - 1u;      /* whole dwords below w, m of dword w */
                    O.l0 = w > 0 ? v[0] : v[0] & m;
                    O.l1 = w > 1 ? v[1] : w == 1 ? v[1] & m : 0u;
                    O.l2 = w > 2 ? v[2] : w == 2 ? v[2] & m : 0u;
                    O.l3 = w == 3 ? v[3] & m : 0u;
                }
                if (O.nseq & 1u) O.q0 = *((const uint64_t *)O.sq - O.nseq);
            }
            QZK_SPROF(3);
            if (j == 0 && h_mode != 0) {
                if (!ok) { seg_status = bad_status < 0 ? bad_status : QZK_INF_ESPEC; seg_done = true; if (!why) why = 25; }
                if (ok && total_out > sg.out_cap) { seg_status = QZK_INF_ESPEC; seg_done = true; why = 30; }   /* the serial kernel says EOUT */
            }
        }
    }
    if (live) qzk_tok_flush(&O);                                    /* what my sub-stream still holds in registers */
#ifdef QZK_SPEC_PROF
    QZK_STAMP_OUT(1);
    QZK_SPROF(3);                                                   /* [3] chain walks and the rest */
    if (j == 0 && live && sidx < (1u << 17)) qzk_spec_seg[sidx] = (unsigned int)((p_segdone ? p_segdone : QZK_PCLK() - p_start) >> 6);
    if (blockIdx.x < 8192) {
        if (lane == 0) { for (int k = 0; k < 3; k++) qzk_spec_prof[blockIdx.x][k] = pacc[k];
                         qzk_spec_prof[blockIdx.x][2] += pacc[3];                   /* (the chain walks: 0.1-0.3 %) */
                         qzk_spec_prof[blockIdx.x][3] = nlook << 40 | (plook & ((1ull << 40) - 1));
                         qzk_spec_prof[blockIdx.x][5] = p_begin << 32 | (__builtin_amdgcn_s_memrealtime() & 0xffffffffull); }   /* begin | end */
        atomicMax(&qzk_spec_prof[blockIdx.x][4], pacc[6]);          /* the lane longest in the hot loop: its clocks there ... */
        atomicMax(&qzk_spec_prof[blockIdx.x][6], pacc[7]);          /* ... the most trips of a lane ... */
        atomicAdd(&qzk_spec_prof[blockIdx.x][7], pacc[7]);          /* ... all lanes' trips */
    }
#endif
    /* ---- lane 0: the segment's result ---- */
    if (live && j == 0) {
        qzk_infres r; r.status = seg_status; r.in_used = 0; r.out_len = 0; r.nblocks = S.nblocks;
        if (seg_status >= 0) { r.in_used = S.b.pos - (uint32_t)(S.b.bc >> 3); r.out_len = total_out; }
        if (r.status < 0) { nel = 0; r.nblocks = why ? why : 40; }                 /* ESPEC: for the serial kernel; anything else: the segment's error */
        C->nel = nel; C->pad = 0;
        res[sidx] = r;
    }
}

#endif
