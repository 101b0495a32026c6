/*
 * qzk_deflate_huff.h — K2: zlib-exact block coding (trees.c behaviour) of the symbol stream K1 produced, ONE WAVE PER
 * CHUNK and no workgroup barriers (qzk_huff_chunk), gfx950 - and the kernels that run it: qzk_lz77_pull_kernel at the
 * end of this file (level 1: the wave that parsed a chunk codes it too, in the same LDS) and qzk_huff_kernel (a launch
 * of its own behind the parse kernels of comp_lvl 2-9).  The workgroup CRC-32 routine (K6) also lives here.
 *
 * Replaces, on the reference's software path, the _tr_flush_block() half of zlib's deflate() (src/qatzip_sw.c:197) and
 * the running crc32 zlib keeps for the gzip trailer; CPU restatement: oracle/qzo_deflate.c (flush_block & co).
 *
 * Per block (<= 32767 symbols): parallel histogram (LDS atomics) -> the three Huffman trees with zlib's exact heap
 * order / tie-break / overflow repair (the heap on lane 0, the loops around it on the wave), the code lengths RLE-coded,
 * stored / fixed / dynamic by zlib's byte-count rule -> 64 lanes at a time turn symbols into bit strings, a wave
 * prefix-sum gives every symbol its bit offset, and the bits are OR-ed into an LDS staging tile that is flushed as whole
 * bytes.  The chunk ends with the Z_FULL_FLUSH marker (000 + pad + 00 00 FF FF) or, for the last chunk of a stream,
 * BFINAL + byte padding.
 */
#ifndef QZK_DEFLATE_HUFF_H
#define QZK_DEFLATE_HUFF_H
#include "qzk_common.h"
#include "qzk_deflate_lz77.h"
#include "qzk_crcmath.h"

#define QZK_HT 256                 /* threads per workgroup */
#define QZK_LCODES 286
#define QZK_DCODES 30
#define QZK_BLCODES 19
#define QZK_HEAP 573

typedef struct { uint32_t tab[4][256]; uint32_t ktab[4][256]; uint32_t x2n[32]; uint32_t red[16]; } qzk_crc_lds;

/* One wave's K2 state, 5.8 KiB: small enough that the sixteen waves of a K1 workgroup can each run K2 on the chunk they
 * just parsed, in the LDS their parse no longer needs (qzk_lz77_pull_kernel below).  Arrays whose lifetimes do not
 * overlap share storage: a tree's codes replace its frequencies (gen_bitlen, their last reader, is done by then), the
 * dynamic header is assembled where the heap stood.  zlib's copy of the frequencies into the tree nodes has no
 * counterpart: a leaf's frequency is read where the histogram left it, an inner node's only ever from its heap entry. */
typedef struct {
    /* frequencies (u32 for LDS atomics) -> code tables: code | len<<16 */
    union { uint32_t fl[288]; uint32_t code_l[288]; };
    union { uint32_t fd[32]; uint32_t code_d[32]; };
    union { uint32_t fbl[20]; uint32_t code_bl[20]; };
    union {
        struct {                            /* tree-build scratch (lane 0) */
            uint32_t heap[QZK_LCODES + 4];  /* freq<<15 | depth<<10 | node: at most one entry per symbol (+ the slot pqdown reads past the end) */
            uint16_t order[QZK_HEAP + 3];   /* zlib keeps this in the back of its heap[] */
        };
        struct { uint32_t hdr[320]; uint32_t hbits; };      /* dynamic header bits */
    };
    uint16_t dad[QZK_HEAP + 3];
    uint8_t len_l[QZK_HEAP + 3], len_d[64], len_bl[40];
    uint16_t bl_count[16];
    /* decisions */
    uint32_t btype, max_l, max_d;
    /* output staging */
    uint32_t stage[104];               /* 64 lanes x 48 bits + the pending partial byte */
} qzk_huff_lds;

QZ_DEV uint32_t qzk_bitrev(uint32_t code, int len)         /* len 1..15 */
{
#ifdef QZ_SIM
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
#else
    return __builtin_bitreverse32(code) >> (32 - len);
#endif
}

/* ------------------------------------------------------------------ lane-0 tree code */
#define QZK_SMALLER(a, b) (((a) >> 10) <= ((b) >> 10))

QZ_DEV void qzk_pqdown(uint32_t *heap, int heap_len, int k)
{
    /* one LDS round trip per TWO levels: the children of k (heap[2k], heap[2k+1]: j is even, the slot after the heap's end
     * exists and is never chosen) and its four grandchildren (heap[4k .. 4k+3]) are asked for together, the smaller child
     * and then the smaller of ITS children are picked in registers - the same comparisons in the same order as zlib's
     * pqdownheap, a level at a time; only the loads are early.  (This loop is a chain of dependent LDS reads on one lane:
     * the serial tree build was a tenth of the fused kernel's time per chunk.) */
    const uint32_t v = heap[k];
    int j = k << 1;
    while (j <= heap_len) {
        const uint32_t c0 = heap[j], c1 = heap[j + 1];
        uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0;
        const bool deep = 2 * j <= heap_len;                       /* a second level exists below either child */
        if (deep) { g0 = heap[2 * j]; g1 = heap[2 * j + 1]; g2 = heap[2 * j + 2]; g3 = heap[2 * j + 3]; }
        uint32_t c = c0; bool right = false;
        if (j < heap_len && QZK_SMALLER(c1, c0)) { j++; c = c1; right = true; }
        if (QZK_SMALLER(v, c)) break;
        heap[k] = c; k = j; j <<= 1;
        if (j > heap_len) break;
        {   /* second level, from the words already here (heap[j], heap[j + 1] with the new j; entries above heap_len are
             * never chosen, exactly as above) */
            const uint32_t d0 = right ? g2 : g0, d1 = right ? g3 : g1;
            uint32_t d = d0;
            if (j < heap_len && QZK_SMALLER(d1, d0)) { j++; d = d1; }
            if (QZK_SMALLER(v, d)) break;
            heap[k] = d; k = j; j <<= 1;
        }
    }
    heap[k] = v;
}

/* zlib build_tree(), taken apart: what is a loop over independent elements runs on the wave, what is the heap stays on
 * lane 0.
 *   qzk_tree_init  (wave)   len = 0 for unused symbols, the symbols in use into heap[1..] in symbol order
 *   qzk_tree_core  (lane 0) forced nodes, heap order, the tree itself, gen_bitlen with the overflow repair, opt / stat
 *   qzk_tree_codes (wave)   gen_codes: a symbol's code is next_code[len] + the number of earlier symbols of that length */
QZ_DEV void qzk_tree_init(qzk_huff_lds *S, const uint32_t *freq, uint8_t *len, int elems, int lane, int *heap_len_out, int *max_code_out)
{
    int heap_len = 0, max_code = -1;
    for (int n0 = 0; n0 < elems; n0 += 64) {
        const int n = n0 + lane;
        const bool in = n < elems;
        const uint32_t f = in ? freq[n] : 0;
        if (in && !f) len[n] = 0;
        const uint64_t nz = qz_ballot(f != 0);
        if (f) S->heap[heap_len + 1 + qz_popc64(nz & qz_below(lane))] = (f << 15) | (uint32_t)n;
        if (nz) max_code = n0 + qz_msb64(nz);
        heap_len += qz_popc64(nz);
    }
    *heap_len_out = heap_len; *max_code_out = max_code;
}

/* lane 0.  Returns max_code.  Adds to *opt / *stat. */
QZ_DEV int qzk_tree_core(qzk_huff_lds *S, uint32_t *freq, uint8_t *len, int elems, int heap_len, int max_code,
                         int max_length, int stype /*0 l,1 d,2 bl*/, uint32_t *opt, uint32_t *stat)
{
    uint32_t *heap = S->heap; uint16_t *order = S->order, *dad = S->dad;
    const uint32_t *nf = freq;              /* leaves only (n <= max_code) */
    int heap_max = QZK_HEAP, n, m, node;

    while (heap_len < 2) {
        node = max_code < 2 ? ++max_code : 0;
        heap[++heap_len] = (1u << 15) | (uint32_t)node;
        freq[node] = 1;
        (*opt)--;
        if (stype == 0) *stat -= (node < 144 ? 8 : node < 256 ? 9 : node < 280 ? 7 : 8);
        else if (stype == 1) *stat -= 5;
    }
    for (n = heap_len / 2; n >= 1; n--) qzk_pqdown(heap, heap_len, n);
    node = elems;
    do {
        uint32_t a = heap[1], b;
        heap[1] = heap[heap_len--];
        qzk_pqdown(heap, heap_len, 1);
        b = heap[1];
        n = (int)(a & 1023); m = (int)(b & 1023);
        order[--heap_max] = (uint16_t)n; order[--heap_max] = (uint16_t)m;
        {
            uint32_t f = (a >> 15) + (b >> 15);
            uint32_t da = (a >> 10) & 31, db = (b >> 10) & 31, d = (da >= db ? da : db) + 1;
            dad[n] = dad[m] = (uint16_t)node;
            heap[1] = (f << 15) | (d << 10) | (uint32_t)node;
        }
        node++;
        qzk_pqdown(heap, heap_len, 1);
    } while (heap_len >= 2);
    order[--heap_max] = (uint16_t)(heap[1] & 1023);

    /* gen_bitlen */
    {
        int h, bits, overflow = 0;
        uint16_t *blc = S->bl_count;
        for (bits = 0; bits <= 15; bits++) blc[bits] = 0;
        len[order[heap_max]] = 0;
        for (h = heap_max + 1; h < QZK_HEAP; h++) {
            int xbits = 0;
            n = order[h];
            bits = len[dad[n]] + 1;
            if (bits > max_length) bits = max_length, overflow++;
            len[n] = (uint8_t)bits;
            if (n > max_code) continue;
            blc[bits]++;
            if (stype == 0) { if (n >= 265 && n < 285) xbits = (n - 261) >> 2; }
            else if (stype == 1) { if (n >= 4) xbits = (n >> 1) - 1; }
            else { xbits = n == 16 ? 2 : n == 17 ? 3 : n == 18 ? 7 : 0; }
            *opt += (uint32_t)nf[n] * (uint32_t)(bits + xbits);
            if (stype == 0) *stat += (uint32_t)nf[n] * (uint32_t)((n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8) + xbits);
            else if (stype == 1) *stat += (uint32_t)nf[n] * (uint32_t)(5 + xbits);
        }
        if (overflow) {
            do {
                bits = max_length - 1;
                while (blc[bits] == 0) bits--;
                blc[bits]--; blc[bits + 1] += 2; blc[max_length]--;
                overflow -= 2;
            } while (overflow > 0);
            for (bits = max_length; bits != 0; bits--) {
                n = blc[bits];
                while (n != 0) {
                    m = order[--h];
                    if (m > max_code) continue;
                    if (len[m] != bits) { *opt += ((uint32_t)bits - len[m]) * nf[m]; len[m] = (uint8_t)bits; }
                    n--;
                }
            }
        }
    }
    return max_code;
}

/* gen_codes on the wave: sixty-four symbols a step; `codes` may be the storage the frequencies were in */
QZ_DEV void qzk_tree_codes(const qzk_huff_lds *S, const uint8_t *len, uint32_t *codes, int max_code, int lane)
{
    uint32_t next_code[16], code = 0;
    for (int bits = 1; bits <= 15; bits++) { code = (code + S->bl_count[bits - 1]) << 1; next_code[bits] = code; }
    for (int n0 = 0; n0 <= max_code; n0 += 64) {
        const int n = n0 + lane;
        const int l = n <= max_code ? len[n] : 0;
        uint32_t mine = 0;
#pragma unroll
        for (int b = 1; b <= 15; b++) {
            const uint64_t m = qz_ballot(l == b);
            if (l == b) mine = next_code[b] + (uint32_t)qz_popc64(m & qz_below(lane));
            next_code[b] += (uint32_t)qz_popc64(m);
        }
        if (n <= max_code) codes[n] = l ? (qzk_bitrev(mine, l) | ((uint32_t)l << 16)) : 0;
    }
}

QZ_DEV void qzk_scan_tree(qzk_huff_lds *S, uint8_t *len, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = n + 1 <= max_code ? len[n + 1] : 0xffff;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) S->fbl[curlen] += (uint32_t)count;
        else if (curlen != 0) { if (curlen != prevlen) S->fbl[curlen]++; S->fbl[16]++; }
        else if (count <= 10) S->fbl[17]++;
        else S->fbl[18]++;
        count = 0; prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}

QZ_DEV void qzk_hdr_bits(qzk_huff_lds *S, uint32_t v, int nb)
{
    uint32_t pos = S->hbits, w = pos >> 5, s = pos & 31;
    S->hdr[w] |= v << s;
    if (s + (uint32_t)nb > 32) S->hdr[w + 1] |= v >> (32 - s);
    S->hbits = pos + (uint32_t)nb;
}
#define QZK_HCODE(S, c) qzk_hdr_bits(S, (S)->code_bl[c] & 0xffff, (int)((S)->code_bl[c] >> 16))

QZ_DEV void qzk_send_tree(qzk_huff_lds *S, uint8_t *len, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = len[0], count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = n + 1 <= max_code ? len[n + 1] : 0xffff;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) { do { QZK_HCODE(S, curlen); } while (--count != 0); }
        else if (curlen != 0) {
            if (curlen != prevlen) { QZK_HCODE(S, curlen); count--; }
            QZK_HCODE(S, 16); qzk_hdr_bits(S, (uint32_t)(count - 3), 2);
        } else if (count <= 10) { QZK_HCODE(S, 17); qzk_hdr_bits(S, (uint32_t)(count - 3), 3); }
        else { QZK_HCODE(S, 18); qzk_hdr_bits(S, (uint32_t)(count - 11), 7); }
        count = 0; prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}

/* the whole wave: everything _tr_flush_block decides.  Leaves S->btype (0 stored,1 fixed,2 dynamic), code tables and
 * (dynamic) header bits, all AFTER the 3-bit block header.  Ends with the wave in step (an LDS sync). */
QZ_DEV void qzk_plan_block(qzk_huff_lds *S, uint32_t stored_len, bool can_store, int lane)
{
    static const uint8_t bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t opt = 0, stat = 0;             /* lane 0's are the ones that count */
    int hl, mc;

    if (lane == 0) S->fl[256] = 1;
    qz_lds_sync();
    qzk_tree_init(S, S->fl, S->len_l, QZK_LCODES, lane, &hl, &mc);
    qz_lds_sync();
    if (lane == 0) S->max_l = (uint32_t)qzk_tree_core(S, S->fl, S->len_l, QZK_LCODES, hl, mc, 15, 0, &opt, &stat);
    qz_lds_sync();
    const int max_l = (int)S->max_l;
    qzk_tree_codes(S, S->len_l, S->code_l, max_l, lane);
    qz_lds_sync();

    qzk_tree_init(S, S->fd, S->len_d, QZK_DCODES, lane, &hl, &mc);
    qz_lds_sync();
    if (lane == 0) S->max_d = (uint32_t)qzk_tree_core(S, S->fd, S->len_d, QZK_DCODES, hl, mc, 15, 1, &opt, &stat);
    qz_lds_sync();
    const int max_d = (int)S->max_d;
    qzk_tree_codes(S, S->len_d, S->code_d, max_d, lane);
    qz_lds_sync();

    if (lane == 0) {
        for (int i = 0; i < 19; i++) S->fbl[i] = 0;
        qzk_scan_tree(S, S->len_l, max_l);
        qzk_scan_tree(S, S->len_d, max_d);
    }
    qz_lds_sync();
    qzk_tree_init(S, S->fbl, S->len_bl, QZK_BLCODES, lane, &hl, &mc);
    qz_lds_sync();
    if (lane == 0) {
        int max_blindex;
        uint32_t opt_lenb, static_lenb;
        qzk_tree_core(S, S->fbl, S->len_bl, QZK_BLCODES, hl, mc, 7, 2, &opt, &stat);
        for (max_blindex = QZK_BLCODES - 1; max_blindex >= 3; max_blindex--)
            if (S->len_bl[bl_order[max_blindex]] != 0) break;
        opt += 3 * ((uint32_t)max_blindex + 1) + 5 + 5 + 4;
        opt_lenb = (opt + 3 + 7) >> 3;
        static_lenb = (stat + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        S->btype = (stored_len + 4 <= opt_lenb && can_store) ? 0u : static_lenb == opt_lenb ? 1u : 2u;
        S->stage[0] = (uint32_t)max_blindex;        /* handed to the header below (the staging tile is idle here) */
    }
    qz_lds_sync();
    const uint32_t btype = S->btype;
    if (btype == 0) return;
    if (btype == 1) {
        /* fixed codes: canonical, lengths 8/9/7/8 and 5 */
        for (int n = lane; n < 288; n += 64) {
            int l = n < 144 ? 8 : n < 256 ? 9 : n < 280 ? 7 : 8;
            uint32_t c = n < 144 ? 0x30 + (uint32_t)n : n < 256 ? 0x190 + (uint32_t)(n - 144)
                         : n < 280 ? (uint32_t)(n - 256) : 0xC0 + (uint32_t)(n - 280);
            S->code_l[n] = qzk_bitrev(c, l) | ((uint32_t)l << 16);
        }
        if (lane < 30) S->code_d[lane] = qzk_bitrev((uint32_t)lane, 5) | (5u << 16);
        qz_lds_sync();
        return;
    }
    const int max_blindex = (int)S->stage[0];
    qz_lds_sync();                                  /* every lane has read it: code_bl may replace fbl, hdr the heap */
    qzk_tree_codes(S, S->len_bl, S->code_bl, QZK_BLCODES - 1, lane);
    for (int i = lane; i < 320; i += 64) S->hdr[i] = 0;
    qz_lds_sync();
    if (lane == 0) {
        S->hbits = 0;
        qzk_hdr_bits(S, (uint32_t)(max_l + 1 - 257), 5);
        qzk_hdr_bits(S, (uint32_t)(max_d + 1 - 1), 5);
        qzk_hdr_bits(S, (uint32_t)(max_blindex + 1 - 4), 4);
        for (int r = 0; r <= max_blindex; r++) qzk_hdr_bits(S, S->len_bl[bl_order[r]], 3);
        qzk_send_tree(S, S->len_l, max_l);
        qzk_send_tree(S, S->len_d, max_d);
    }
    qz_lds_sync();
}

/* ------------------------------------------------------------------ wave helpers */
QZ_DEV uint32_t qzk_wave_incl_scan(uint32_t v, int lane) { (void)lane; return qz_wave_incl_scan(v); }

/* symbol -> (bits, nbits) with the current code tables */
QZ_DEV void qzk_sym_bits(const qzk_huff_lds *S, uint32_t lc, uint32_t dist, uint64_t *val, uint32_t *nb)
{
    if (dist == 0) {
        uint32_t c = S->code_l[lc];
        *val = c & 0xffff; *nb = c >> 16;
        return;
    }
    uint32_t lcode, lext, lval, dcode, dext, dval, D = dist - 1;
    if (lc < 8) { lcode = lc; lext = 0; lval = 0; }
    else if (lc == 255) { lcode = 28; lext = 0; lval = 0; }
    else {
        uint32_t k = 31 - (uint32_t)__builtin_clz(lc);
        lcode = 4 * (k - 1) + ((lc >> (k - 2)) & 3); lext = k - 2; lval = lc & ((1u << lext) - 1);
    }
    if (D < 4) { dcode = D; dext = 0; dval = 0; }
    else {
        uint32_t k = 31 - (uint32_t)__builtin_clz(D);
        dcode = 2 * k + ((D >> (k - 1)) & 1); dext = k - 1; dval = D & ((1u << dext) - 1);
    }
    uint32_t cl = S->code_l[257 + lcode], cd = S->code_d[dcode];
    uint64_t v = cl & 0xffff; uint32_t n = cl >> 16;
    v |= (uint64_t)lval << n; n += lext;
    v |= (uint64_t)(cd & 0xffff) << n; n += cd >> 16;
    v |= (uint64_t)dval << n; n += dext;
    *val = v; *nb = n;
}

typedef struct { uint8_t *out; uint32_t nbytes; uint32_t cbits; uint32_t carry; } qzk_bitout;   /* wave-uniform */

/* One wave-wide emission step: every lane contributes (val, nb <= 48 bits), in lane order.  The bits are OR-ed into
 * an LDS tile behind the pending partial byte and the whole bytes leave as (unaligned) dwords. */
QZ_DEV void qzk_emit_wave(qzk_huff_lds *S, qzk_bitout *bo, uint64_t val, uint32_t nb, int lane)
{
    const uint32_t inc = qzk_wave_incl_scan(nb, lane);
    const uint32_t total = qz_readlane(inc, 63);
    if (total == 0) return;
    const uint32_t tb = bo->cbits + total, nw = (tb + 31) >> 5;           /* bits / words in the tile after this step */
    for (uint32_t i = (uint32_t)lane; i < nw + 2; i += 64) S->stage[i] = i == 0 ? bo->carry : 0;
    qz_lds_sync();
    if (nb) {
        const uint32_t pos = bo->cbits + inc - nb, w = pos >> 5, sh = pos & 31;
        const uint64_t a = val << sh;
        const uint32_t hi = sh ? (uint32_t)(val >> (64 - sh)) : 0;
        atomicOr(&S->stage[w], (uint32_t)a);
        if ((uint32_t)(a >> 32)) atomicOr(&S->stage[w + 1], (uint32_t)(a >> 32));
        if (hi) atomicOr(&S->stage[w + 2], hi);
    }
    qz_lds_sync();
    const uint32_t nby = tb >> 3;
    uint8_t *o = bo->out + bo->nbytes;
    for (uint32_t i = 4u * (uint32_t)lane; i < nby; i += 256) {
        const uint32_t v = S->stage[i >> 2];
        if (i + 4 <= nby) ((qz_u32u *)(o + i))->v = v;
        else for (uint32_t k = 0; i + k < nby; k++) o[i + k] = (uint8_t)(v >> (8 * k));
    }
    const uint32_t nc = tb & 7;
    const uint32_t cw = (S->stage[nby >> 2] >> (8 * (nby & 3))) & ((1u << nc) - 1);
    qz_lds_sync();                                                        /* the tile is rewritten by the next step */
    bo->nbytes += nby; bo->cbits = nc; bo->carry = qz_readfirstlane(cw);
}

/* flush the partial byte (bi_windup) */
QZ_DEV void qzk_align(qzk_bitout *bo, int lane)
{
    if (bo->cbits) {
        if (lane == 0) bo->out[bo->nbytes] = (uint8_t)bo->carry;
        bo->nbytes++; bo->cbits = 0; bo->carry = 0;
    }
}

/* ------------------------------------------------------------------ CRC32 (K6) */
/* crc32 of src[0..n) by the whole workgroup (finalised, zlib convention).
 * Layout: the 256 threads sweep the data in rows of 4 KiB, thread t owning the 16 bytes at row*4096 + 16*t, so
 * every wave load is one coalesced 1 KiB request.  A thread folds its pieces Horner-style:
 *     acc = acc * x^(8*4096) ^ crc32(piece)          (crc32_combine algebra; the constant multiply is 4 LDS lookups)
 * then shifts acc by the bytes that follow its last piece, and the partial results are XOR-reduced. */
QZ_DEV uint32_t qzk_crc16(const qzk_crc_lds *S, uint32_t a, uint32_t b, uint32_t c2, uint32_t d)
{
    uint32_t c = 0xffffffffu ^ a;
    c = S->tab[3][c & 0xff] ^ S->tab[2][(c >> 8) & 0xff] ^ S->tab[1][(c >> 16) & 0xff] ^ S->tab[0][c >> 24]; c ^= b;
    c = S->tab[3][c & 0xff] ^ S->tab[2][(c >> 8) & 0xff] ^ S->tab[1][(c >> 16) & 0xff] ^ S->tab[0][c >> 24]; c ^= c2;
    c = S->tab[3][c & 0xff] ^ S->tab[2][(c >> 8) & 0xff] ^ S->tab[1][(c >> 16) & 0xff] ^ S->tab[0][c >> 24]; c ^= d;
    c = S->tab[3][c & 0xff] ^ S->tab[2][(c >> 8) & 0xff] ^ S->tab[1][(c >> 16) & 0xff] ^ S->tab[0][c >> 24];
    return ~c;
}

QZ_DEV uint32_t qzk_block_crc32(qzk_crc_lds *S, const uint8_t *src, uint32_t n)
{
    const int t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    {
        uint32_t c = (uint32_t)t;
        for (int k = 0; k < 8; k++) c = (c & 1) ? QZK_POLY ^ (c >> 1) : c >> 1;
        S->tab[0][t] = c;
    }
    if (t == 0) {
        uint32_t p = 1u << 30;
        S->x2n[0] = p;
        for (int i = 1; i < 32; i++) S->x2n[i] = p = qzk_multmodp(p, p);
    }
    qz_block_sync();
    {
        uint32_t c0 = S->tab[0][t], c1, c2, c3;
        c1 = (c0 >> 8) ^ S->tab[0][c0 & 0xff];
        c2 = (c1 >> 8) ^ S->tab[0][c1 & 0xff];
        c3 = (c2 >> 8) ^ S->tab[0][c2 & 0xff];
        S->tab[1][t] = c1; S->tab[2][t] = c2; S->tab[3][t] = c3;
        /* ktab[k][b] = (b << 8k) * x^(8*4096) mod P : multiply-by-constant as four byte lookups */
        const uint32_t K = qzk_x2nmodp(S->x2n, 4096, 3);
        for (int k = 0; k < 4; k++) S->ktab[k][t] = qzk_multmodp(K, (uint32_t)t << (8 * k));
    }
    qz_block_sync();
    const uint32_t rows = n >> 12;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rows; r++) {
        const uint8_t *q = src + ((uint64_t)r << 12) + 16u * (uint32_t)t;
        const uint32_t a = qz_ld32(q), b = qz_ld32(q + 4), c = qz_ld32(q + 8), d = qz_ld32(q + 12);
        acc = S->ktab[0][acc & 0xff] ^ S->ktab[1][(acc >> 8) & 0xff] ^ S->ktab[2][(acc >> 16) & 0xff] ^ S->ktab[3][acc >> 24];
        acc ^= qzk_crc16(S, a, b, c, d);
    }
    uint32_t part = 0;
    if (rows) {
        const uint32_t tail = n - ((rows - 1) << 12) - 16u * (uint32_t)t - 16u;   /* bytes after my last full-row piece */
        part = qzk_multmodp(qzk_x2nmodp(S->x2n, tail, 3), acc);
    }
    {   /* the remaining n mod 4096 bytes: one piece of <= 16 bytes per thread */
        const uint32_t o = (rows << 12) + 16u * (uint32_t)t;
        if (o < n) {
            const uint32_t len = n - o < 16 ? n - o : 16;
            uint32_t c = 0xffffffffu;
            for (uint32_t i = 0; i < len; i++) c = S->tab[0][(c ^ src[o + i]) & 0xff] ^ (c >> 8);
            c = ~c;
            const uint32_t tail = n - o - len;
            part ^= tail ? qzk_multmodp(qzk_x2nmodp(S->x2n, tail, 3), c) : c;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) part ^= qz_shfl(part, lane ^ d);
    if (lane == 0) S->red[wv] = part;
    qz_block_sync();
    uint32_t r = 0;
    for (int k = 0; k < QZK_HT / 64; k++) r ^= S->red[k];
    qz_block_sync();
    return r;
}

/* ------------------------------------------------------------------ the kernel */
#define QZK_HW 64                  /* threads per workgroup of K2: one wave per chunk, no workgroup barriers */

/* symbols written to HBM by the SAME wave a moment ago (K1 and K2 fused): read them at the L2 (no copy of these lines
 * can be in this CU's L1 - they were never read before in this launch - but the load then does not depend on that) */
template <bool L2> QZ_DEV uint32_t qzk_sym_ld8(const uint8_t *p)
{
#ifndef QZ_SIM
    if (L2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return *p;
}
template <bool L2> QZ_DEV uint32_t qzk_sym_ld16(const uint16_t *p)
{
#ifndef QZ_SIM
    if (L2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return *p;
}

/* K2 for one chunk by one wave: `in` = the chunk's input, lcs/dists/mt = what K1 left for it, `out` = the chunk's slot */
template <bool L2>
QZ_DEV void qzk_huff_chunk(qzk_huff_lds *Sp, const int lane, const uint8_t *in, const uint8_t *lcs, const uint16_t *dists,
                           const qzk_lzmeta *mt, uint8_t *out, const bool is_final, uint32_t *out_len, uint64_t *prof_plan = 0)
{
    qzk_huff_lds &S = *Sp;
    (void)prof_plan;
    const uint32_t n = mt->n, nsym = mt->nsym, nfull = mt->nfull, cs = mt->can_store;

    qzk_bitout bo;
    bo.out = out; bo.nbytes = 0; bo.cbits = 0; bo.carry = 0;
    /* blocks 0..nfull-1 are full; block nfull is the remainder (possibly empty) */
    const uint32_t nblocks = nfull + ((is_final || nsym > nfull * QZK_LITBUF) ? 1 : 0);
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t s0 = b * QZK_LITBUF;
        const uint32_t s1 = b < nfull ? s0 + QZK_LITBUF : nsym;
        const uint32_t bs = mt->bstart[b < QZK_MAXBLK ? b : QZK_MAXBLK - 1];
        const uint32_t be = b < nfull ? mt->bstart[b + 1 < QZK_MAXBLK ? b + 1 : QZK_MAXBLK - 1] : n;
        const uint32_t last = (is_final && b == nblocks - 1) ? 1u : 0u;

        for (int i = lane; i < 288; i += QZK_HW) S.fl[i] = 0;
        if (lane < 32) S.fd[lane] = 0;
        qz_lds_sync();
        for (uint32_t i = s0 + (uint32_t)lane; i < s1; i += QZK_HW) {
            uint32_t lc = qzk_sym_ld8<L2>(lcs + i), dist = qzk_sym_ld16<L2>(dists + i);
            if (dist == 0) atomicAdd(&S.fl[lc], 1u);
            else {
                uint32_t lcode, D = dist - 1, dcode;
                if (lc < 8) lcode = lc; else if (lc == 255) lcode = 28;
                else { uint32_t k = 31 - (uint32_t)__builtin_clz(lc); lcode = 4 * (k - 1) + ((lc >> (k - 2)) & 3); }
                if (D < 4) dcode = D; else { uint32_t k = 31 - (uint32_t)__builtin_clz(D); dcode = 2 * k + ((D >> (k - 1)) & 1); }
                atomicAdd(&S.fl[257 + lcode], 1u);
                atomicAdd(&S.fd[dcode], 1u);
            }
        }
        qz_lds_sync();
#if defined(QZK_PROF) && !defined(QZ_SIM)
        const uint64_t tp_ = __builtin_readcyclecounter();
#endif
        /* zlib's trees: the heap is one lane following a chain of LDS round trips while the CU's other waves parse, the
         * loops around it run on the wave.  At raised priority its instructions go out as soon as their operands are there
         * instead of waiting for a turn among the other waves of the SIMD */
#ifndef QZ_SIM
        __builtin_amdgcn_s_setprio(3);
#endif
        qzk_plan_block(&S, be - bs, (cs >> b) & 1, lane);
#ifndef QZ_SIM
        __builtin_amdgcn_s_setprio(0);
#endif
#if defined(QZK_PROF) && !defined(QZ_SIM)
        if (prof_plan) *prof_plan += __builtin_readcyclecounter() - tp_;
#endif
        const uint32_t btype = S.btype;

        /* 3-bit block header */
        qzk_emit_wave(&S, &bo, lane == 0 ? (uint64_t)((btype << 1) | last) : 0, lane == 0 ? 3 : 0, lane);

        if (btype == 0) {
            const uint32_t slen = be - bs;
            qzk_align(&bo, lane);
            if (lane < 4) {
                uint32_t v = lane < 2 ? slen : ~slen;
                bo.out[bo.nbytes + (uint32_t)lane] = (uint8_t)(v >> (8 * (lane & 1)));
            }
            for (uint32_t i = (uint32_t)lane; i < slen; i += QZK_HW) bo.out[bo.nbytes + 4 + i] = in[bs + i];
            bo.nbytes += 4 + slen;
        } else {
            if (btype == 2) {
                const uint32_t hb = S.hbits, hw = (hb + 31) >> 5;
                for (uint32_t w0 = 0; w0 < hw; w0 += QZK_HW) {
                    uint32_t w = w0 + (uint32_t)lane, nb = 0; uint64_t v = 0;
                    if (w < hw) { nb = (w + 1) * 32 <= hb ? 32 : hb - w * 32; v = S.hdr[w] & (nb == 32 ? 0xffffffffu : ((1u << nb) - 1)); }
                    qzk_emit_wave(&S, &bo, v, nb, lane);
                }
            }
            for (uint32_t i0 = s0; i0 < s1; i0 += QZK_HW) {
                uint32_t i = i0 + (uint32_t)lane, nb = 0; uint64_t v = 0;
                if (i < s1) qzk_sym_bits(&S, qzk_sym_ld8<L2>(lcs + i), qzk_sym_ld16<L2>(dists + i), &v, &nb);
                qzk_emit_wave(&S, &bo, v, nb, lane);
            }
            {   /* END_BLOCK */
                uint32_t c = S.code_l[256];
                qzk_emit_wave(&S, &bo, lane == 0 ? (uint64_t)(c & 0xffff) : 0, lane == 0 ? c >> 16 : 0, lane);
            }
        }
        qz_lds_sync();
    }
    if (is_final) qzk_align(&bo, lane);
    else {
        /* Z_FULL_FLUSH: empty stored block */
        qzk_emit_wave(&S, &bo, 0, lane == 0 ? 3 : 0, lane);
        qzk_align(&bo, lane);
        if (lane < 4) bo.out[bo.nbytes + (uint32_t)lane] = lane < 2 ? 0x00 : 0xff;
        bo.nbytes += 4;
    }
    *out_len = bo.nbytes;          /* wave-uniform: every lane stores the same word (no lane-0-only block at the end of a
                                    * pull-loop iteration, see qzk_lz77_pull_kernel) */
}

QZ_KERNEL_MAX(64) qzk_huff_kernel(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                          const uint8_t *sym_lc, const uint16_t *sym_dist, const qzk_lzmeta *meta,
                          uint8_t *slots, uint32_t slot_stride, uint32_t final_chunk /* index or ~0u */,
                          uint32_t *out_len, const uint32_t *cdesc)
{
    QZ_LDS qzk_huff_lds S;
    const uint32_t chunk = blockIdx.x;
    if (chunk >= nchunks) return;
    const uint64_t coff = (uint64_t)chunk * chunk_sz;
    const bool is_final = cdesc ? (cdesc[chunk] & QZK_CDESC_FINAL) != 0 : chunk == final_chunk;
    (void)src_len;
    qzk_huff_chunk<false>(&S, qz_lane(), src + coff, sym_lc + coff, sym_dist + coff, meta + chunk,
                          slots + (uint64_t)chunk * slot_stride, is_final, out_len + chunk);
}

/* ------------------------------------------------------------------ K1 (+ K2) launch shape
 * Persistent workgroups of QZK_K1_WAVES waves, QZK_K1_OCC per CU; every WAVE pulls chunk numbers from a counter (uneven chunks
 * balance themselves) and owns one column of its workgroup's table: entry h of wave w at
 * tables[(row group * 65536 + h) * QZK_K1_TABW + place * QZK_K1_WAVES + w] (qzk_deflate_lz77.h).  The waves never talk to each other - what they share is cache
 * lines.  epoch_base + chunk number = the chunk's epoch (host: unique per chunk across launches, never 0).
 *
 * With `slots` the wave that parsed a chunk also codes it (K2, qzk_huff_chunk) before it pulls the next one, in the LDS
 * its parse no longer needs.  K2's serial tree build is one lane following a chain of LDS round trips, the parse a wave
 * following a chain of HBM round trips: as a separate launch beside the next batch's K1, K2 (two waves per CU in the LDS
 * K1 leaves) stretched that K1 launch by 7 of its 23 ms; inside the K1 waves it fills issue slots the parse leaves
 * empty, and a chunk's symbols are read back while they are still on their way through the L2. */
/* Input that is still crossing PCIe when the launch starts (qzd_deflate_raw_from_host): `avail` points at two words of
 * pinned HOST memory.  avail[0] = chunks of this launch whose bytes have landed in HBM - the host raises it as the
 * pieces of its copy complete (a copy engine moves them, no compute unit is needed for it; these workgroups fill the
 * register files, so nothing else could run beside them anyway).  A wave that has pulled chunk k waits until chunk k+1
 * has landed too (the ring reads a few hundred bytes past the window, never 1 KiB), asleep for about as long as the
 * missing chunks take on the link, then drops its L1: the lines it is about to read have never been touched by this
 * launch, so no L2 holds an older copy of them.  A wave that waits for a second in vain (the host is gone) says so in
 * avail[1] and leaves; 0xffffffff in avail[0] is the host giving up. */
QZ_DEV bool qzk_wait_input(const uint32_t *avail, uint32_t need, uint32_t chunk_sz)
{
#ifdef QZ_SIM
    (void)avail; (void)need; (void)chunk_sz;
    return true;
#else
    const uint32_t per = chunk_sz >> 14 ? chunk_sz >> 14 : 1u;             /* naps of ~0.27 us: about 1 us per missing 64 KiB */
    uint32_t slept = 0;
    for (;;) {
        const uint32_t wm = qz_readfirstlane(__hip_atomic_load(avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (wm == 0xffffffffu) return false;
        if (wm >= need) break;
        const uint32_t naps = (need - wm < 4096u ? need - wm : 4096u) * per;
        for (uint32_t k = 0; k < naps; k++) __builtin_amdgcn_s_sleep(10);
        slept += naps;
        if (slept > (4u << 20)) {                                           /* more than a second asleep */
            __hip_atomic_store((uint32_t *)avail + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
#endif
}

/* ------------------------------------------------------------------ the stream leaves from inside the launch
 * When one launch covers the whole call, the wave that coded a chunk also moves its bytes from the chunk's slot to their
 * place in the destination - HBM, or the caller's pinned buffer across PCIe, which then carries the stream while the
 * parse is still running instead of for 7 ms after it (1 GiB call).  A chunk's place is the sum of the lengths before
 * it, so:
 *   pub[c]   stream length of chunk c + 1, stored (agent scope) by the wave that coded it; 0 = not yet
 *   front    chunks << 40 | bytes: every chunk below `chunks` has its offset; bytes = their total
 *   offs[c]  offset of chunk c + 1, written by whoever moved the front past c
 * Any wave that has just published a length tries to move the front: sixty-four lanes look at pub[front ..], the run of
 * coded chunks from the front on gets its offsets from one wave prefix sum, one atomic maximum publishes the new front
 * (a wave that was overtaken has computed and stored the very same offsets).  A wave copies its own chunks only (bytes it stored itself, so
 * its own L2 has them - no fence), oldest first, as soon as the front has passed them; what is still behind an unfinished
 * older chunk waits in a list of eight (LDS) while the wave parses on, and is waited for when the list is full or the
 * chunks have run out.  The wave that moves the front to the end stores the total.  overflow: 1 = destination too small,
 * 2 = a wave waited two seconds in vain (some other wave is gone). */
typedef struct {
    uint8_t *dst; uint64_t cap;          /* dst NULL: a scan and a gather kernel behind the launch do it */
    uint64_t *front; uint32_t *pub; uint64_t *offs;
    uint64_t *running; uint32_t *overflow;
} qzk_outp;
#define QZK_OUT_PEND 8

#ifndef QZ_SIM
QZ_DEV uint64_t qzk_out_advance(const qzk_outp O, uint32_t nchunks, int lane)
{
    for (;;) {
        const uint64_t f0 = __hip_atomic_load(O.front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t f = (uint64_t)qz_readfirstlane((uint32_t)f0) | (uint64_t)qz_readfirstlane((uint32_t)(f0 >> 32)) << 32;
        const uint32_t F = (uint32_t)(f >> 40);
        const uint64_t B = f & ((1ull << 40) - 1);
        if (F >= nchunks) return f;
        const uint32_t idx = F + (uint32_t)lane;
        const uint32_t p = idx < nchunks ? __hip_atomic_load(O.pub + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint64_t ok = qz_ballot(p != 0);
        const int k = ~ok ? qz_ctz64(~ok) : 64;
        if (k == 0) return f;
        const uint32_t l = lane < k ? p - 1 : 0u;
        const uint32_t incl = qzk_wave_incl_scan(l, lane);
        if (lane < k) __hip_atomic_store(O.offs + idx, B + incl - l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t tot = qz_readlane(incl, 63);
        const uint64_t nf = (uint64_t)(F + (uint32_t)k) << 40 | (B + tot);
        /* the front only grows and a later front is a larger word: a maximum publishes it, whoever else got there first.
         * A wave-uniform operand: the compiler issues ONE atomic for the wave, as for the chunk counter */
        __hip_atomic_fetch_max(O.front, nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (F + (uint32_t)k >= nchunks) *O.running = B + tot;          /* every wave that gets here stores the same total */
    }
}

/* copy out what the front has passed; block: until nothing is pending (ph == pt on return unless a wave is gone) */
QZ_DEV void qzk_out_drain(const qzk_outp O, const uint8_t *slots, uint32_t slot_stride, const uint32_t *pend, uint32_t *ph,
                          uint32_t pt, uint32_t nchunks, bool block, int lane)
{
    uint32_t naps = 0;
    while (*ph != pt) {
        const uint64_t f = qzk_out_advance(O, nchunks, lane);
        const uint32_t c0 = pend[*ph % QZK_OUT_PEND];
        if ((uint32_t)(f >> 40) > c0) {
            uint64_t o1 = 0;
            for (uint32_t t = 0; t < (1u << 20) && o1 == 0; t++) {                     /* stored before the front moved; a few turns at worst */
                const uint64_t v = __hip_atomic_load(O.offs + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                o1 = (uint64_t)qz_readfirstlane((uint32_t)v) | (uint64_t)qz_readfirstlane((uint32_t)(v >> 32)) << 32;
            }
            const uint32_t n = qz_readfirstlane(__hip_atomic_load(O.pub + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - 1;
            const uint64_t o = o1 - 1;
            if (o1 == 0) atomicOr(O.overflow, 2u);
            else if (o + n > O.cap) atomicOr(O.overflow, 1u);
            else {
                const uint8_t *s = slots + (uint64_t)c0 * slot_stride;
                uint8_t *d = O.dst + o;
                const uint32_t nw = n >> 2;
                uint32_t i = (uint32_t)lane;
                for (; i + 192 < nw; i += 256) {                                        /* four loads in flight per lane */
                    const uint32_t a = ((const uint32_t *)s)[i], b = ((const uint32_t *)s)[i + 64], c = ((const uint32_t *)s)[i + 128],
                                   e = ((const uint32_t *)s)[i + 192];
                    ((qz_u32u *)d)[i].v = a; ((qz_u32u *)d)[i + 64].v = b; ((qz_u32u *)d)[i + 128].v = c; ((qz_u32u *)d)[i + 192].v = e;
                }
                for (; i < nw; i += 64) ((qz_u32u *)d)[i].v = ((const uint32_t *)s)[i];
                for (uint32_t j = (nw << 2) + (uint32_t)lane; j < n; j += 64) d[j] = s[j];
            }
            (*ph)++;
            continue;
        }
        if (!block) return;
        __builtin_amdgcn_s_sleep(38);
        if (++naps > (2u << 20)) { atomicOr(O.overflow, 2u); return; }
    }
}
#endif

#define QZK_K1_LDSW (QZK_K1_PARSEW > (sizeof(qzk_huff_lds) + 3) / 4 ? QZK_K1_PARSEW : (sizeof(qzk_huff_lds) + 3) / 4)
/* FED: the launch may start before its input is there (avail) and may move the coded chunks to the destination itself
 * (O.dst) - the form calls fed from host memory take; the resident form is compiled without either, so that what it
 * never uses costs it no registers */
template <bool FED>
QZ_DEV void qzk_lz77_pull_body(const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks,
                                                     uint8_t *sym_lc, uint16_t *sym_dist, qzk_lzmeta *meta, qzk_bkt *tables,
                                                     uint32_t *counter, const uint32_t *cdesc, uint32_t epoch_base,
                                                     uint8_t *slots, uint32_t slot_stride, uint32_t final_chunk, uint32_t *out_len,
                                                     uint32_t *crc_out /* per chunk, or NULL */,
                                                     const uint32_t *avail /* NULL: the input is all there */, const qzk_outp O)
{
    QZ_LDS uint32_t lds_all[QZK_K1_WAVES][QZK_K1_LDSW + QZK_OUT_PEND];
    QZ_LDS qzk_k1crc_lds crcT;
    if (crc_out) qzk_k1crc_init(&crcT);             /* kernel argument: the whole workgroup takes the same way */
    const int wv = (int)(threadIdx.x >> 6);
    uint32_t *const lds = lds_all[wv];
    qzk_bkt *tab = tables + ((size_t)((blockIdx.x / (8u * QZK_K1_TABGRP)) * 8u + (blockIdx.x & 7u)) * QZK_HSIZE) * QZK_K1_TABW
                          + ((blockIdx.x >> 3) % QZK_K1_TABGRP) * QZK_K1_WAVES + wv;
    uint32_t *const pend = lds + QZK_K1_LDSW;       /* chunks coded by this wave whose bytes have not left their slots yet */
    uint32_t ph = 0, pt = 0;
    for (;;) {
        /* no `if (lane == 0)` block at the start or the end of this loop's body: the compiler threads lane-0-only blocks
         * of consecutive iterations together, after which the other 63 lanes would run readfirstlane without lane 0 (seen
         * on gfx950: the wave re-parses chunk 0 forever).  Every lane takes part; only lane 0 adds. */
        uint32_t chunk = atomicAdd(counter, qz_lane() == 0 ? 1u : 0u);
        chunk = qz_readfirstlane(chunk);
        if (chunk >= nchunks) break;
        if (FED && avail) {                            /* kernel argument, wave-uniform values */
            const uint32_t ahead = 1u + 1024u / chunk_sz;                   /* whole chunks the ring's read-ahead may touch */
            if (!qzk_wait_input(avail, chunk + 1 + ahead < nchunks ? chunk + 1 + ahead : nchunks, chunk_sz)) break;
        }
        /* symbols: with K2 in the wave they only live until the wave has coded them - one chunk's worth per WAVE (read
         * back at the L2: the same addresses carried the previous chunk's symbols); without, one per chunk of the launch */
        const uint64_t soff = slots ? (uint64_t)(blockIdx.x * QZK_K1_WAVES + (uint32_t)wv) * chunk_sz : (uint64_t)chunk * chunk_sz;
        uint8_t *const wlc = sym_lc + soff;
        uint16_t *const wdist = sym_dist + soff;
        qzk_lz77_chunk(src, src_len, chunk_sz, chunk, wlc, wdist, meta, tab, epoch_base + chunk, cdesc, lds,
                       crc_out ? &crcT : (const qzk_k1crc_lds *)0, crc_out ? crc_out + chunk : (uint32_t *)0);
        if (slots) {
            /* every lane stored the chunk's meta words itself; the symbols were stored by other lanes of this wave:
             * memory operations of one wave reach the L2 in issue order */
            qz_lds_sync();
            const bool is_final = cdesc ? (cdesc[chunk] & QZK_CDESC_FINAL) != 0 : chunk == final_chunk;
#if defined(QZK_K1_NOK2)                    /* measurement only (profiles/r5_k1_without_k2.txt): the parse and the CRC, the symbols are dropped */
            (void)is_final;
            out_len[chunk] = 0;                     /* wave-uniform: every lane stores the same word */
#elif defined(QZK_PROF) && !defined(QZ_SIM)
            const uint64_t tk_ = __builtin_readcyclecounter();
            uint64_t plan_ = 0;
            qzk_huff_chunk<true>((qzk_huff_lds *)lds, qz_lane(), src + (uint64_t)chunk * chunk_sz, wlc, wdist, meta + chunk,
                                 slots + (uint64_t)chunk * slot_stride, is_final, out_len + chunk, &plan_);
            meta[chunk].prof[13] = __builtin_readcyclecounter() - tk_; meta[chunk].prof[14] = plan_;
#else
            qzk_huff_chunk<true>((qzk_huff_lds *)lds, qz_lane(), src + (uint64_t)chunk * chunk_sz, wlc, wdist, meta + chunk,
                                 slots + (uint64_t)chunk * slot_stride, is_final, out_len + chunk);
#endif
            qz_lds_sync();
#ifndef QZ_SIM
            if (FED && O.dst) {                     /* kernel argument */
                const uint32_t n1 = qz_readfirstlane(__hip_atomic_load(out_len + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1;
                __hip_atomic_store(O.pub + chunk, n1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* every lane, the same word */
                pend[pt % QZK_OUT_PEND] = chunk;
                pt++;
                qz_lds_sync();
                qzk_out_drain(O, slots, slot_stride, pend, &ph, pt, nchunks, pt - ph == QZK_OUT_PEND, qz_lane());
            }
#endif
        }
    }
#ifndef QZ_SIM
    if (FED && O.dst && slots) qzk_out_drain(O, slots, slot_stride, pend, &ph, pt, nchunks, true, qz_lane());
#endif
}
#define QZK_PULL_PARAMS const uint8_t *src, uint64_t src_len, uint32_t chunk_sz, uint32_t nchunks, uint8_t *sym_lc, uint16_t *sym_dist, \
    qzk_lzmeta *meta, qzk_bkt *tables, uint32_t *counter, const uint32_t *cdesc, uint32_t epoch_base, uint8_t *slots, \
    uint32_t slot_stride, uint32_t final_chunk, uint32_t *out_len, uint32_t *crc_out, const uint32_t *avail, const qzk_outp O
#define QZK_PULL_ARGS src, src_len, chunk_sz, nchunks, sym_lc, sym_dist, meta, tables, counter, cdesc, epoch_base, slots, slot_stride, \
    final_chunk, out_len, crc_out, avail, O
QZ_KERNEL_OCC(64 * QZK_K1_WAVES, (QZK_K1_OCC * QZK_K1_WAVES + 3) / 4) qzk_lz77_pull_kernel(QZK_PULL_PARAMS) { qzk_lz77_pull_body<false>(QZK_PULL_ARGS); }
QZ_KERNEL_OCC(64 * QZK_K1_WAVES, (QZK_K1_OCC * QZK_K1_WAVES + 3) / 4) qzk_lz77_pull_fed_kernel(QZK_PULL_PARAMS) { qzk_lz77_pull_body<true>(QZK_PULL_ARGS); }

#endif
