/*
 * qzk_inflate_spec.h — K3b phase A with SUB-SEGMENT SPECULATION: K lanes per segment, gfx950.
 *
 * Phase A of the two-phase inflate (qzk_inflate_lane.h) is bound by a lone lane's serial
 * walk over its segment: 1.4 us per symbol whatever the segment count.  Huffman streams
 * resynchronise on their own: a decoder started at an arbitrary bit falls into step with
 * the true symbol boundaries after a few symbols.  So a segment is decoded by a GROUP of K
 * lanes, block by block ("rounds"):
 *   1. lane 0 of the group parses the block header where the previous round ended (stored
 *      blocks it copies itself, they are a memcpy) and builds the Huffman tables, which
 *      the group shares in LDS;
 *   2. the K lanes start at K evenly spaced bit offsets of what is left of the segment and
 *      decode into their own literal / sequence sub-streams.  A lane that reaches its
 *      neighbour's start keeps going until its own symbol boundary coincides with one the
 *      neighbour published (every 8th of the neighbour's first 128 symbol starts, in LDS):
 *      from there on the two decode identically, so the lane stops and the neighbour's
 *      tokens take over from that record.  Lane 0 is right by construction, hence by
 *      induction every lane on the chain is.  A neighbour that never falls into step (or
 *      ran into garbage) is skipped: the lane carries on to the next one;
 *   3. lane 0 walks the chain, appends the pieces to the segment's piece list (what phase B,
 *      qzk_lz_resolve_kernel, stitches together) and takes over the bit position behind the
 *      block's END_BLOCK for the next round.
 * Whatever does not fit (bad data, a sub-stream outgrowing its scratch, too many pieces) is
 * answered with QZK_INF_ESPEC and the host decodes that segment with the serial kernel:
 * this kernel either delivers a fully validated segment or nothing.
 * Same place in the reference as the rest of K3: zlib inflate(), src/qatzip_sw.c:339.
 */
#ifndef QZK_INFLATE_SPEC_H
#define QZK_INFLATE_SPEC_H
#include "qzk_inflate_lane.h"

#define QZK_SPEC_NREC 16           /* symbol starts a lane publishes per round ... */
#define QZK_SPEC_EVERY 8           /* ... one every so many symbols (covers the first 128) */
#define QZK_SPEC_REACH 8u          /* a lane gives up 1/REACH of a share beyond its own share (it should have fallen into step) */
#define QZK_SPEC_MINBITS 512u      /* a block is split only when every lane gets at least this much of it */
/* scratch of one sub-stream: QZK_SPEC_SLACK times its fair share of the segment's worst case */
#define QZK_SPEC_SLACK 4ull
#define QZK_SPEC_LITCAP(out_cap, K) (((QZK_SPEC_SLACK * (uint64_t)(out_cap) / (K) + 255) & ~(uint64_t)63) + 64)
#define QZK_SPEC_SEQCAP(out_cap, K) ((QZK_SPEC_SLACK * ((uint64_t)(out_cap) / 3) / (K) + 24) & ~(uint64_t)1)

typedef struct { uint32_t nlit, nseq, lrun, olen; } qzk_rec;      /* token-stream counters at a published symbol start */
/* what lane 0 tells its group before a round: mode 0 = segment finished, 1 = decode this block */
typedef struct { uint32_t mode, hdr_end, span, last; int lmax, dmax; } qzk_spec_hdr;
enum { QZK_ST_RUN = 0, QZK_ST_SYNC, QZK_ST_EOB, QZK_ST_REDO };
/* how a lane's round ended: SYNC with (target, cidx), or EOB (its block ended at bit `at`), or REDO */
typedef struct { int kind; uint32_t target, cidx, at, nlit0, nseq0, lrun0, olen0, nlit, nseq, olen; } qzk_spec_stop;

template <int K>
QZ_KERNEL_MAX(64) qzk_inflate_spec_kernel(const uint8_t *comp, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs,
                                  qzk_inf_tab *tabs, const qzk_tokseg *ts /* [nsegs * K] */, uint8_t *lits, qzk_seq *seqs,
                                  qzk_chain *chains, qzk_rec *recs /* [nsegs * K * QZK_SPEC_NREC] */)
{
    constexpr int SPW = 64 / K;                                     /* segments per wave */
    QZ_LDS uint16_t roots[SPW][QZK_LANE_ROOTSZ];
    QZ_LDS qzk_spec_hdr ghdr[SPW];
    QZ_LDS qzk_spec_stop gstop[64];
    QZ_LDS uint32_t gpos[64][QZK_SPEC_NREC];                        /* published symbol starts (bit offsets) */
    QZ_LDS uint32_t gnrec[64];

    const int lane = (int)threadIdx.x, g = lane / K, j = lane % K;
    const uint32_t sidx = blockIdx.x * SPW + (uint32_t)g;
    const bool live = sidx < nsegs;
    const qzk_infseg sg = segs[live ? sidx : 0];
    qzk_inf_tab *T = tabs + (live ? sidx : 0);
    uint16_t *const lroot = roots[g], *const droot = lroot + (1 << QZK_LLROOT);
    const uint32_t slot = (live ? sidx : 0) * K + (uint32_t)j;     /* my sub-stream */
    qzk_rec *myrec = recs + (uint64_t)slot * QZK_SPEC_NREC;
    qzk_chain *C = chains + (live ? sidx : 0);

    qzk_lane_st S;
    S.b.p = comp + sg.in_off; S.b.end = sg.in_len; qzk_lseek(&S.b, 0);
    S.op = 0; S.nblocks = 0; S.last = 0; S.clen = 0; S.rpos = 0;
    S.out_cap = j == 0 ? sg.out_cap : 0xffffffffu;                 /* lane 0 knows the output offset, phase B checks the rest */
    S.lmax = 0; S.dmax = 0; S.lbase = 0; S.status = QZK_INF_EDATA; S.state = QZK_LS_HDR;
    S.through = sg.flags & QZK_INF_THROUGH_FLUSH;
    qzk_tok_out O;
    qzk_tok_init(&O, lits + ts[slot].lit_off, seqs + ts[slot].seq_off, false);
    const uint32_t lit_cap = (uint32_t)QZK_SPEC_LITCAP(sg.out_cap, K) - 64, seq_cap = (uint32_t)QZK_SPEC_SEQCAP(sg.out_cap, K) - 10;
    const uint32_t limit_bits = 8u * (sg.pad != 0 && sg.pad < sg.in_len ? sg.pad : sg.in_len);

    /* lane 0 only: the segment's result so far */
    uint32_t nel = 0, total_out = 0, blk_bits = 0;                  /* pieces, output bytes, length of the previous Huffman block */
    int seg_status = QZK_INF_ESPEC;                                 /* set to FINAL / FLUSH when the segment ends well */
    uint32_t why = 0;                                               /* developer aid: why the segment was handed back */
    bool seg_done = !live;                                          /* lane 0: nothing more to do for this segment */

    for (;;) {
        /* ---- 1. lane 0: block headers and stored blocks up to the next Huffman block ---- */
        if (j == 0) {
            qzk_spec_hdr H; H.mode = 0; H.hdr_end = 0; H.span = 0; H.last = 0; H.lmax = 0; H.dmax = 0;
            while (!seg_done && H.mode == 0) {
                S.op = total_out;
                qzk_lane_header(&S, T, lroot, droot);
                if (S.state == QZK_LS_RAW) {
                    /* stored block: nothing to decode - phase B copies it straight from the input */
                    if (nel >= QZK_CHAIN_MAXEL) { seg_done = true; break; }
                    qzk_chain_el e; e.sub = QZK_PIECE_RAW; e.seq_first = S.rpos; e.seq_count = S.clen; e.lit_first = 0; e.lrun_skip = 0;
                    C->el[nel++] = e;
                    total_out += S.clen; S.clen = 0;
                    if (S.last) { seg_status = QZK_INF_FINAL; seg_done = true; }
                    else S.state = QZK_LS_HDR;
                } else if (S.state == QZK_LS_SYM) {
                    const uint32_t at = 8u * S.b.pos - (uint32_t)S.b.bc;
                    H.mode = 1; H.hdr_end = at; H.last = S.last; H.lmax = S.lmax; H.dmax = S.dmax;
                    /* what the group shares: the rest of the segment, or - from the second block on - a block as long as
                     * the previous one (zlib closes a block every 32767 symbols, so blocks of a segment are alike).
                     * Too little to share: span 0 parks the other lanes, lane 0 decodes alone */
                    uint32_t share = limit_bits > at ? limit_bits - at : 0;
                    /* more than 5 bits per output byte: mostly literals, the segment will take several 32767-symbol
                     * blocks - guess half of it for the first one */
                    if (!blk_bits && (uint64_t)limit_bits > 5ull * sg.out_cap) blk_bits = share / 2;
                    if (blk_bits && blk_bits + blk_bits / 8 < share) share = blk_bits + blk_bits / 8;
                    H.span = share > QZK_SPEC_MINBITS * K ? share / K : 0;
                } else if (S.state == QZK_LS_HDR) {
                    /* an empty stored block in through-mode: next header */
                } else {                                            /* DONE: FINAL / FLUSH, or an error for the serial kernel to name */
                    if (S.status == QZK_INF_FINAL || S.status == QZK_INF_FLUSH) seg_status = S.status;
                    seg_done = true;
                }
            }
            ghdr[g] = H;
        }
        gnrec[lane] = 0;
        qz_lds_sync();
        const qzk_spec_hdr H = ghdr[g];
        if (qz_ballot(H.mode != 0) == 0) break;                     /* every group of the wave has finished */

        /* ---- 2. the group decodes the block ---- */
        bool active = H.mode != 0 && (j == 0 || H.span != 0);
        if (active && j > 0) {                                      /* my guessed start */
            const uint32_t at = H.hdr_end + (uint32_t)j * H.span;
            qzk_lseek(&S.b, at >> 3);
            qzk_lrefill(&S.b);
            QZK_DROP(&S.b, at & 7);
            S.last = H.last; S.lmax = H.lmax; S.dmax = H.dmax; S.op = 0;
        }
        S.state = QZK_LS_SYM;
        /* a lane that started beyond the end of the block (the block was shorter than guessed) decodes garbage and never
         * falls into step: it gives up shortly after its own share, lane 0 (always right) never does */
        const uint32_t give_up = j == 0 || H.span == 0 ? 0xffffffffu
                               : j == K - 1 ? H.hdr_end + ((uint32_t)K + 2) * H.span          /* the block may be longer than guessed */
                               : H.hdr_end + ((uint32_t)j + 1) * H.span + H.span / QZK_SPEC_REACH;
        uint32_t nsym = 0, target = H.span ? (uint32_t)j + 1 : (uint32_t)K, cursor = 0;
        qzk_spec_stop st; st.kind = QZK_ST_RUN; st.target = 0; st.cidx = 0; st.at = 0;
        st.nlit0 = QZK_NLIT(O) - O.lrun; st.nseq0 = O.nseq; st.lrun0 = 0; st.olen0 = S.op;   /* where my piece of this round starts */

        /* what every symbol start does before the symbol: publish / look for the neighbour's footprint / watch the scratch */
#define QZK_SPEC_PRE() do { \
        const uint32_t at_ = 8u * S.b.pos - (uint32_t)S.b.bc; \
        if ((nsym % QZK_SPEC_EVERY) == 0 && nsym < QZK_SPEC_EVERY * QZK_SPEC_NREC) { \
            const uint32_t k_ = nsym / QZK_SPEC_EVERY; \
            qzk_rec r_; r_.nlit = QZK_NLIT(O); r_.nseq = O.nseq; r_.lrun = O.lrun; r_.olen = S.op; \
            myrec[k_] = r_; gpos[lane][k_] = at_; gnrec[lane] = k_ + 1; \
        } \
        nsym++; \
        while (st.kind == QZK_ST_RUN && target < (uint32_t)K && at_ >= H.hdr_end + target * H.span) { \
            const int tl_ = lane - j + (int)target; const uint32_t tn_ = gnrec[tl_]; \
            while (cursor < tn_ && gpos[tl_][cursor] < at_) cursor++; \
            if (cursor < tn_ && gpos[tl_][cursor] == at_) { st.kind = QZK_ST_SYNC; st.target = target; st.cidx = cursor; } \
            else if (cursor >= tn_) { target++; cursor = 0; }       /* never fell into step: carry on to the next one */ \
            else break; \
        } \
        if (st.kind == QZK_ST_RUN && (QZK_NLIT(O) > lit_cap || O.nseq > seq_cap || at_ >= give_up)) { \
            st.kind = QZK_ST_REDO; st.cidx = at_ >= give_up ? 1u : 2u; } \
    } while (0)

        while (qz_ballot(active) != 0) {
            if (active) {
                qzk_lbits *b = &S.b;
                if (b->pos + 16 <= b->end) {
                    b->pos -= (uint32_t)(b->bc >> 3); b->bc &= 7; b->bb &= (1ull << b->bc) - 1;
                    uint64_t pw = qzk_ld64u(b->p + b->pos);
                    for (int trip = 0; trip < 256 && st.kind == QZK_ST_RUN && S.state == QZK_LS_SYM && b->pos + 16 <= b->end; trip++) {
                        QZK_SPEC_PRE();
                        if (st.kind == QZK_ST_RUN) {
                            b->bb |= pw << b->bc;
                            b->pos += (uint32_t)(63 - b->bc) >> 3; b->bc |= 56;
                            pw = qzk_ld64u(b->p + b->pos);
                            qzk_lane_symbol<false, false>(&S, &O, T, lroot, droot, (uint64_t)1 << 40);
                            qzk_tok_round_flush(&O);        /* one symbol a trip here: whole words leave as they fill */
                        }
                    }
                    b->pos -= (uint32_t)(b->bc >> 3); b->bc &= 7; b->bb &= (1ull << b->bc) - 1;
                    const uint32_t keep = (uint32_t)b->bc; const uint64_t low = b->bb;
                    qzk_lseek(b, b->pos);
                    b->bb = low; b->bc = (int)keep;
                } else {
                    for (int trip = 0; trip < 64 && st.kind == QZK_ST_RUN && S.state == QZK_LS_SYM; trip++) {
                        QZK_SPEC_PRE();
                        if (st.kind == QZK_ST_RUN) {
                            qzk_lrefill(b);
                            qzk_lane_symbol<true, false>(&S, &O, T, lroot, droot, (uint64_t)1 << 40);
                            qzk_tok_round_flush(&O);
                        }
                    }
                }
                if (st.kind == QZK_ST_RUN && S.state != QZK_LS_SYM) {
                    /* END_BLOCK leaves HDR (or DONE with FINAL when the block was the last one); anything else is an error */
                    if (S.state == QZK_LS_HDR || (S.state == QZK_LS_DONE && S.status == QZK_INF_FINAL)) {
                        st.kind = QZK_ST_EOB; st.at = 8u * b->pos - (uint32_t)b->bc;
                    } else { st.kind = QZK_ST_REDO; st.cidx = 3u; }
                }
                if (st.kind != QZK_ST_RUN) {
                    if (O.lrun) qzk_tok_seq(&O, 0u, 0u);            /* my piece ends with its pending literals */
                    qzk_tok_flush(&O);
                    st.nlit = QZK_NLIT(O); st.nseq = O.nseq; st.olen = S.op;
                    gstop[lane] = st;
                    active = false;
                }
            }
        }
#undef QZK_SPEC_PRE
        qz_wave_sync();
        /* ---- 3. lane 0 walks the chain of this round ---- */
        if (j == 0 && H.mode != 0) {
            uint32_t cur = 0, c_nlit = 0, c_nseq = 0, c_lrun = 0, c_olen = 0;
            bool from_rec = false, ok = false;
            for (int hop = 0; hop < K; hop++) {
                const qzk_spec_stop s = gstop[lane + (int)cur];
                if (s.kind != QZK_ST_SYNC && s.kind != QZK_ST_EOB) { why = 10 + s.cidx; break; }
                if (nel >= QZK_CHAIN_MAXEL) { why = 20; break; }
                qzk_chain_el e; e.sub = cur;
                if (from_rec) { e.seq_first = c_nseq; e.lit_first = c_nlit; e.lrun_skip = c_lrun; total_out += s.olen - c_olen; }
                else { e.seq_first = s.nseq0; e.lit_first = s.nlit0; e.lrun_skip = 0; total_out += s.olen - s.olen0; }
                e.seq_count = s.nseq - e.seq_first;
                C->el[nel++] = e;
                if (s.kind == QZK_ST_EOB) {
                    ok = true; blk_bits = s.at - H.hdr_end;
                    qzk_lseek(&S.b, s.at >> 3); qzk_lrefill(&S.b); QZK_DROP(&S.b, s.at & 7);   /* I carry on behind the block */
                    if (H.last) { seg_status = QZK_INF_FINAL; seg_done = true; } else S.state = QZK_LS_HDR;
                    break;
                }
                const qzk_rec q = recs[((uint64_t)sidx * K + s.target) * QZK_SPEC_NREC + s.cidx];
                cur = s.target; from_rec = true; c_nlit = q.nlit; c_nseq = q.nseq; c_lrun = q.lrun; c_olen = q.olen;
            }
            if (!ok) { seg_status = QZK_INF_ESPEC; seg_done = true; }
            if (ok && total_out > sg.out_cap) { seg_status = QZK_INF_ESPEC; seg_done = true; why = 30; }   /* the serial kernel says EOUT */
        }
    }
    /* ---- lane 0: the segment's result ---- */
    if (live && j == 0) {
        qzk_infres r; r.status = seg_status; r.in_used = 0; r.out_len = 0; r.nblocks = S.nblocks;
        if (seg_status >= 0) {
            if (O.lrun && nel < QZK_CHAIN_MAXEL) {                  /* stored bytes at the very end: a piece of their own */
                qzk_chain_el e; e.sub = 0; e.seq_first = O.nseq; e.seq_count = 1; e.lit_first = QZK_NLIT(O) - O.lrun; e.lrun_skip = 0;
                C->el[nel++] = e;
                qzk_tok_seq(&O, 0u, 0u);
            } else if (O.lrun) r.status = QZK_INF_ESPEC;
            qzk_tok_flush(&O);
            r.in_used = S.b.pos - (uint32_t)(S.b.bc >> 3); r.out_len = total_out;
        }
        if (r.status < 0) { r.status = QZK_INF_ESPEC; nel = 0; r.nblocks = why ? why : 40; }
        C->nel = nel; C->pad = 0;
        res[sidx] = r;
    }
}

#endif
