/*
 * qzo_deflate.c — restatement of zlib 1.2.11's deflate for the configuration
 * the QATzip software path fixes:  deflateInit2(level, Z_DEFLATED, +-15/31,
 * MAX_MEM_LEVEL(9), Z_DEFAULT_STRATEGY)   (src/qatzip_sw.c:147-152), levels 1-3
 * (zlib's greedy "deflate_fast" family) and 4-9 (lazy "deflate_slow"), driven
 * per hw_buff_sz chunk with Z_FULL_FLUSH / Z_FINISH (src/qatzip_sw.c:178-231).
 * TEST INFRASTRUCTURE (see qzo.h).
 *
 * zlib is not vendored in /root/reference; this file restates its published
 * algorithm (RFC 1951 + the documented heuristics of deflate.c / trees.c):
 *   - 64 KiB sliding window, 16-bit rolling hash ((h<<6)^c)&0xffff, head/prev
 *     chains, NIL==0, MAX_DIST = 32768-262, greedy parse without lazy eval,
 *     max_insert_length / nice_length / max_chain per level;
 *   - blocks cut after 32767 symbols (lit_bufsize-1 at memLevel 9);
 *   - heap-built Huffman trees with (freq,depth) tie-break, bit-length overflow
 *     repair, RLE'd code-length tree, stored/fixed/dynamic choice.
 * Each chunk is compressed from a fresh state: Z_FULL_FLUSH byte-aligns the
 * stream and clears the hash, so the bytes equal what the continuous stream
 * emits (checked in tests/test_oracle.py against libz).
 */
#include "qzo.h"
#include <stdlib.h>
#include <string.h>

#define WSIZE      32768u
#define WMASK      32767u
#define WINDOW_SZ  65536u
#define MIN_MATCH  3
#define MAX_MATCH  258
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)
#define MAX_DIST   (WSIZE - MIN_LOOKAHEAD)
#define LIT_BUFSIZE 32768u
#define L_CODES 286
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SIZE (2 * L_CODES + 1)
#define END_BLOCK 256
#define REP_3_6 16
#define REPZ_3_10 17
#define REPZ_11_138 18

static const int extra_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const int extra_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const int extra_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

typedef struct { uint16_t freq, code, dad, len; } node_t;

typedef struct {
    node_t *tree; const node_t *stree; const int *extra; int base, elems, max_length, max_code;
} tdesc_t;

static node_t static_ltree[L_CODES + 2], static_dtree[D_CODES];
static uint8_t length_code[256], dist_code[512];
static int base_length[29], base_dist[30];
static int tables_ready;

typedef struct {
    /* input */
    const uint8_t *in; uint32_t avail_in;
    /* LZ77 state (zlib names) */
    uint8_t window[WINDOW_SZ + 8];
    uint16_t head[65536], prev[WSIZE];
    uint32_t strstart, lookahead, match_start, match_length, ins_h;
    long block_start;
    int max_chain, nice_match, max_insert;          /* max_insert doubles as max_lazy_match, as in zlib */
    int good_match, lazy;                           /* lazy: levels 4-9 */
    uint32_t prev_length, prev_match; int match_available;
    /* symbol buffers */
    uint16_t d_buf[LIT_BUFSIZE]; uint8_t l_buf[LIT_BUFSIZE]; uint32_t last_lit;
    /* trees */
    node_t dyn_ltree[HEAP_SIZE], dyn_dtree[2 * D_CODES + 1], bl_tree[2 * BL_CODES + 1];
    tdesc_t l_desc, d_desc, bl_desc;
    uint16_t bl_count[16];
    int heap[HEAP_SIZE], heap_len, heap_max; uint8_t depth[HEAP_SIZE];
    uint32_t opt_len, static_len;
    /* bit writer */
    uint8_t *out; size_t out_pos, out_cap; int overflow;
    uint64_t bi_buf; int bi_valid;
    /* optional symbol dump */
    uint8_t *dump_lc; uint16_t *dump_dist; size_t dump_n, dump_cap;
} dstate_t;

static unsigned bi_reverse(unsigned code, int len)
{
    unsigned r = 0;
    do { r |= code & 1; code >>= 1; r <<= 1; } while (--len > 0);
    return r >> 1;
}

static void gen_codes(node_t *tree, int max_code, const uint16_t *bl_count)
{
    uint16_t next_code[16]; unsigned code = 0;
    for (int bits = 1; bits <= 15; bits++) {
        code = (code + bl_count[bits - 1]) << 1;
        next_code[bits] = (uint16_t)code;
    }
    for (int n = 0; n <= max_code; n++) {
        int len = tree[n].len;
        if (len == 0) continue;
        tree[n].code = (uint16_t)bi_reverse(next_code[len]++, len);
    }
}

static void tables_init(void)
{
    int n, code, length = 0, dist = 0;
    uint16_t bl_count[16] = {0};
    for (code = 0; code < 28; code++) {
        base_length[code] = length;
        for (n = 0; n < (1 << extra_lbits[code]); n++) length_code[length++] = (uint8_t)code;
    }
    length_code[length - 1] = (uint8_t)code;       /* 258 -> code 285 */
    base_length[28] = 0;
    for (code = 0; code < 16; code++) {
        base_dist[code] = dist;
        for (n = 0; n < (1 << extra_dbits[code]); n++) dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < D_CODES; code++) {
        base_dist[code] = dist << 7;
        for (n = 0; n < (1 << (extra_dbits[code] - 7)); n++) dist_code[256 + dist++] = (uint8_t)code;
    }
    n = 0;
    while (n <= 143) static_ltree[n++].len = 8, bl_count[8]++;
    while (n <= 255) static_ltree[n++].len = 9, bl_count[9]++;
    while (n <= 279) static_ltree[n++].len = 7, bl_count[7]++;
    while (n <= 287) static_ltree[n++].len = 8, bl_count[8]++;
    gen_codes(static_ltree, L_CODES + 1, bl_count);
    for (n = 0; n < D_CODES; n++) {
        static_dtree[n].len = 5;
        static_dtree[n].code = (uint16_t)bi_reverse((unsigned)n, 5);
    }
    tables_ready = 1;
}

#define D_CODE(d) ((d) < 256 ? dist_code[d] : dist_code[256 + ((d) >> 7)])

/* ---------------- bit writer ---------------- */
static void put_byte(dstate_t *s, unsigned b)
{
    if (s->out_pos < s->out_cap) s->out[s->out_pos++] = (uint8_t)b; else s->overflow = 1;
}
static void send_bits(dstate_t *s, unsigned value, int length)
{
    s->bi_buf |= (uint64_t)value << s->bi_valid;
    s->bi_valid += length;
    while (s->bi_valid >= 8) { put_byte(s, (unsigned)(s->bi_buf & 0xff)); s->bi_buf >>= 8; s->bi_valid -= 8; }
}
static void bi_windup(dstate_t *s)
{
    if (s->bi_valid > 0) put_byte(s, (unsigned)(s->bi_buf & 0xff));
    s->bi_buf = 0; s->bi_valid = 0;
}
#define send_code(s, c, tree) send_bits(s, (tree)[c].code, (tree)[c].len)

/* ---------------- trees ---------------- */
static void init_block(dstate_t *s)
{
    int n;
    for (n = 0; n < L_CODES; n++) s->dyn_ltree[n].freq = 0;
    for (n = 0; n < D_CODES; n++) s->dyn_dtree[n].freq = 0;
    for (n = 0; n < BL_CODES; n++) s->bl_tree[n].freq = 0;
    s->dyn_ltree[END_BLOCK].freq = 1;
    s->opt_len = s->static_len = 0;
    s->last_lit = 0;
}

#define SMALLER(tree, n, m) \
    ((tree)[n].freq < (tree)[m].freq || ((tree)[n].freq == (tree)[m].freq && s->depth[n] <= s->depth[m]))

static void pqdownheap(dstate_t *s, node_t *tree, int k)
{
    int v = s->heap[k], j = k << 1;
    while (j <= s->heap_len) {
        if (j < s->heap_len && SMALLER(tree, s->heap[j + 1], s->heap[j])) j++;
        if (SMALLER(tree, v, s->heap[j])) break;
        s->heap[k] = s->heap[j]; k = j; j <<= 1;
    }
    s->heap[k] = v;
}

static void gen_bitlen(dstate_t *s, tdesc_t *desc)
{
    node_t *tree = desc->tree; int max_code = desc->max_code;
    const node_t *stree = desc->stree; const int *extra = desc->extra;
    int base = desc->base, max_length = desc->max_length;
    int h, n, m, bits, xbits, overflow = 0; unsigned f;

    for (bits = 0; bits <= 15; bits++) s->bl_count[bits] = 0;
    tree[s->heap[s->heap_max]].len = 0;
    for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
        n = s->heap[h];
        bits = tree[tree[n].dad].len + 1;
        if (bits > max_length) bits = max_length, overflow++;
        tree[n].len = (uint16_t)bits;
        if (n > max_code) continue;
        s->bl_count[bits]++;
        xbits = 0;
        if (n >= base) xbits = extra[n - base];
        f = tree[n].freq;
        s->opt_len += f * (unsigned)(bits + xbits);
        if (stree) s->static_len += f * (unsigned)(stree[n].len + xbits);
    }
    if (overflow == 0) return;
    do {
        bits = max_length - 1;
        while (s->bl_count[bits] == 0) bits--;
        s->bl_count[bits]--;
        s->bl_count[bits + 1] += 2;
        s->bl_count[max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (bits = max_length; bits != 0; bits--) {
        n = s->bl_count[bits];
        while (n != 0) {
            m = s->heap[--h];
            if (m > max_code) continue;
            if ((unsigned)tree[m].len != (unsigned)bits) {
                s->opt_len += ((unsigned)bits - tree[m].len) * tree[m].freq;
                tree[m].len = (uint16_t)bits;
            }
            n--;
        }
    }
}

static void build_tree(dstate_t *s, tdesc_t *desc)
{
    node_t *tree = desc->tree; const node_t *stree = desc->stree;
    int elems = desc->elems, n, m, max_code = -1, node;

    s->heap_len = 0; s->heap_max = HEAP_SIZE;
    for (n = 0; n < elems; n++) {
        if (tree[n].freq != 0) { s->heap[++s->heap_len] = max_code = n; s->depth[n] = 0; }
        else tree[n].len = 0;
    }
    while (s->heap_len < 2) {
        node = s->heap[++s->heap_len] = (max_code < 2 ? ++max_code : 0);
        tree[node].freq = 1; s->depth[node] = 0; s->opt_len--;
        if (stree) s->static_len -= stree[node].len;
    }
    desc->max_code = max_code;
    for (n = s->heap_len / 2; n >= 1; n--) pqdownheap(s, tree, n);
    node = elems;
    do {
        n = s->heap[1]; s->heap[1] = s->heap[s->heap_len--]; pqdownheap(s, tree, 1);
        m = s->heap[1];
        s->heap[--s->heap_max] = n; s->heap[--s->heap_max] = m;
        tree[node].freq = (uint16_t)(tree[n].freq + tree[m].freq);
        s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        tree[n].dad = tree[m].dad = (uint16_t)node;
        s->heap[1] = node++;
        pqdownheap(s, tree, 1);
    } while (s->heap_len >= 2);
    s->heap[--s->heap_max] = s->heap[1];
    gen_bitlen(s, desc);
    gen_codes(tree, max_code, s->bl_count);
}

static void scan_tree(dstate_t *s, node_t *tree, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    tree[max_code + 1].len = 0xffff;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = tree[n + 1].len;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) s->bl_tree[curlen].freq += (uint16_t)count;
        else if (curlen != 0) { if (curlen != prevlen) s->bl_tree[curlen].freq++; s->bl_tree[REP_3_6].freq++; }
        else if (count <= 10) s->bl_tree[REPZ_3_10].freq++;
        else s->bl_tree[REPZ_11_138].freq++;
        count = 0; prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}

static void send_tree(dstate_t *s, node_t *tree, int max_code)
{
    int n, prevlen = -1, curlen, nextlen = tree[0].len, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    for (n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = tree[n + 1].len;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) { do { send_code(s, curlen, s->bl_tree); } while (--count != 0); }
        else if (curlen != 0) {
            if (curlen != prevlen) { send_code(s, curlen, s->bl_tree); count--; }
            send_code(s, REP_3_6, s->bl_tree); send_bits(s, (unsigned)(count - 3), 2);
        } else if (count <= 10) { send_code(s, REPZ_3_10, s->bl_tree); send_bits(s, (unsigned)(count - 3), 3); }
        else { send_code(s, REPZ_11_138, s->bl_tree); send_bits(s, (unsigned)(count - 11), 7); }
        count = 0; prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}

static int build_bl_tree(dstate_t *s)
{
    int max_blindex;
    scan_tree(s, s->dyn_ltree, s->l_desc.max_code);
    scan_tree(s, s->dyn_dtree, s->d_desc.max_code);
    build_tree(s, &s->bl_desc);
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
        if (s->bl_tree[bl_order[max_blindex]].len != 0) break;
    s->opt_len += 3 * ((unsigned)max_blindex + 1) + 5 + 5 + 4;
    return max_blindex;
}

static void compress_block(dstate_t *s, const node_t *ltree, const node_t *dtree)
{
    unsigned dist, lc, code, lx = 0; int extra;
    if (s->last_lit != 0) do {
        dist = s->d_buf[lx]; lc = s->l_buf[lx++];
        if (dist == 0) { send_code(s, lc, ltree); }
        else {
            code = length_code[lc];
            send_code(s, code + 257, ltree);
            extra = extra_lbits[code];
            if (extra) { lc -= (unsigned)base_length[code]; send_bits(s, lc, extra); }
            dist--;
            code = D_CODE(dist);
            send_code(s, code, dtree);
            extra = extra_dbits[code];
            if (extra) { dist -= (unsigned)base_dist[code]; send_bits(s, dist, extra); }
        }
    } while (lx < s->last_lit);
    send_code(s, END_BLOCK, ltree);
}

static void stored_block(dstate_t *s, const uint8_t *buf, uint32_t stored_len, int last)
{
    send_bits(s, (0u << 1) + (unsigned)last, 3);
    bi_windup(s);
    put_byte(s, stored_len & 0xff); put_byte(s, (stored_len >> 8) & 0xff);
    put_byte(s, ~stored_len & 0xff); put_byte(s, (~stored_len >> 8) & 0xff);
    for (uint32_t i = 0; i < stored_len; i++) put_byte(s, buf[i]);
}

static void flush_block(dstate_t *s, int last)
{
    const uint8_t *buf = s->block_start >= 0 ? &s->window[s->block_start] : NULL;
    uint32_t stored_len = (uint32_t)((long)s->strstart - s->block_start);
    uint32_t opt_lenb, static_lenb; int max_blindex;

    build_tree(s, &s->l_desc);
    build_tree(s, &s->d_desc);
    max_blindex = build_bl_tree(s);
    opt_lenb = (s->opt_len + 3 + 7) >> 3;
    static_lenb = (s->static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;

    if (stored_len + 4 <= opt_lenb && buf != NULL) {
        stored_block(s, buf, stored_len, last);
    } else if (static_lenb == opt_lenb) {
        send_bits(s, (1u << 1) + (unsigned)last, 3);
        compress_block(s, static_ltree, static_dtree);
    } else {
        int lcodes = s->l_desc.max_code + 1, dcodes = s->d_desc.max_code + 1, blcodes = max_blindex + 1;
        send_bits(s, (2u << 1) + (unsigned)last, 3);
        send_bits(s, (unsigned)(lcodes - 257), 5);
        send_bits(s, (unsigned)(dcodes - 1), 5);
        send_bits(s, (unsigned)(blcodes - 4), 4);
        for (int rank = 0; rank < blcodes; rank++) send_bits(s, s->bl_tree[bl_order[rank]].len, 3);
        send_tree(s, s->dyn_ltree, lcodes - 1);
        send_tree(s, s->dyn_dtree, dcodes - 1);
        compress_block(s, s->dyn_ltree, s->dyn_dtree);
    }
    init_block(s);
    if (last) bi_windup(s);
    s->block_start = (long)s->strstart;
}

static int tally(dstate_t *s, unsigned dist, unsigned lc)
{
    if (s->dump_lc && s->dump_n < s->dump_cap) {
        s->dump_lc[s->dump_n] = (uint8_t)lc; s->dump_dist[s->dump_n] = (uint16_t)dist;
    }
    s->dump_n++;
    s->d_buf[s->last_lit] = (uint16_t)dist;
    s->l_buf[s->last_lit++] = (uint8_t)lc;
    if (dist == 0) s->dyn_ltree[lc].freq++;
    else {
        dist--;
        s->dyn_ltree[length_code[lc] + 257].freq++;
        s->dyn_dtree[D_CODE(dist)].freq++;
    }
    return s->last_lit == LIT_BUFSIZE - 1;
}

/* ---------------- LZ77 (deflate_fast) ---------------- */
#define UPDATE_HASH(h, c) (h = (((h) << 6) ^ (c)) & 0xffff)

static void fill_window(dstate_t *s)
{
    uint32_t more, n;
    do {
        more = WINDOW_SZ - s->lookahead - s->strstart;
        if (s->strstart >= WSIZE + MAX_DIST) {
            memcpy(s->window, s->window + WSIZE, WSIZE - more);
            s->match_start -= WSIZE; s->strstart -= WSIZE; s->block_start -= (long)WSIZE;
            for (n = 0; n < 65536; n++) s->head[n] = (uint16_t)(s->head[n] >= WSIZE ? s->head[n] - WSIZE : 0);
            for (n = 0; n < WSIZE; n++) s->prev[n] = (uint16_t)(s->prev[n] >= WSIZE ? s->prev[n] - WSIZE : 0);
            more += WSIZE;
        }
        if (s->avail_in == 0) break;
        n = s->avail_in < more ? s->avail_in : more;
        memcpy(s->window + s->strstart + s->lookahead, s->in, n);
        s->in += n; s->avail_in -= n; s->lookahead += n;
        if (s->lookahead >= MIN_MATCH) {            /* insert == 0 in this usage */
            s->ins_h = s->window[s->strstart];
            UPDATE_HASH(s->ins_h, s->window[s->strstart + 1]);
        }
    } while (s->lookahead < MIN_LOOKAHEAD && s->avail_in != 0);
}

static uint32_t longest_match(dstate_t *s, uint32_t cur_match)
{
    unsigned chain_length = (unsigned)s->max_chain;
    const uint8_t *scan = s->window + s->strstart, *match;
    int len, best_len = (int)s->prev_length, nice_match = s->nice_match;    /* prev_length stays 2 in the greedy levels */
    uint32_t limit = s->strstart > MAX_DIST ? s->strstart - MAX_DIST : 0;

    if (s->prev_length >= (uint32_t)s->good_match) chain_length >>= 2;     /* a good match already: search less */
    if ((uint32_t)nice_match > s->lookahead) nice_match = (int)s->lookahead;
    do {
        match = s->window + cur_match;
        /* common-prefix length, at most MAX_MATCH, never past the valid data
         * (zlib compares into stale bytes but clamps; equivalent) */
        int maxlen = s->lookahead < MAX_MATCH ? (int)s->lookahead : MAX_MATCH;
        len = 0;
        while (len < maxlen && match[len] == scan[len]) len++;
        if (len > best_len) {
            s->match_start = cur_match; best_len = len;
            if (len >= nice_match) break;
        }
    } while ((cur_match = s->prev[cur_match & WMASK]) > limit && --chain_length != 0);
    return (uint32_t)best_len <= s->lookahead ? (uint32_t)best_len : s->lookahead;
}

#define INSERT_STRING(s, str, mh) \
    (UPDATE_HASH((s)->ins_h, (s)->window[(str) + 2]), \
     mh = (s)->prev[(str) & WMASK] = (s)->head[(s)->ins_h], (s)->head[(s)->ins_h] = (uint16_t)(str))

static void deflate_fast_chunk(dstate_t *s, int final)
{
    uint32_t hash_head; int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead == 0) break;
        }
        hash_head = 0;
        if (s->lookahead >= MIN_MATCH) INSERT_STRING(s, s->strstart, hash_head);
        if (hash_head != 0 && s->strstart - hash_head <= MAX_DIST)
            s->match_length = longest_match(s, hash_head);
        if (s->match_length >= MIN_MATCH) {
            bflush = tally(s, s->strstart - s->match_start, s->match_length - MIN_MATCH);
            s->lookahead -= s->match_length;
            if (s->match_length <= (uint32_t)s->max_insert && s->lookahead >= MIN_MATCH) {
                s->match_length--;
                do { s->strstart++; INSERT_STRING(s, s->strstart, hash_head); } while (--s->match_length != 0);
                s->strstart++;
            } else {
                s->strstart += s->match_length; s->match_length = 0;
                s->ins_h = s->window[s->strstart];
                UPDATE_HASH(s->ins_h, s->window[s->strstart + 1]);
            }
        } else {
            bflush = tally(s, 0, s->window[s->strstart]);
            s->lookahead--; s->strstart++;
        }
        if (bflush) flush_block(s, 0);
    }
    if (final) { flush_block(s, 1); return; }
    if (s->last_lit) flush_block(s, 0);
    /* Z_FULL_FLUSH: empty stored block, byte aligned */
    stored_block(s, NULL, 0, 0);
}

/* levels 4-9: lazy evaluation.  A match found at p is held back one step; it is emitted only if the search at p+1
 * (shortened when the held match is already good, skipped when it reaches max_lazy) finds nothing longer, otherwise
 * byte p goes out as a literal and the new match is held in turn.  Every position of an emitted match is inserted. */
#define TOO_FAR 4096
static void deflate_slow_chunk(dstate_t *s, int final)
{
    uint32_t hash_head; int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead == 0) break;
        }
        hash_head = 0;
        if (s->lookahead >= MIN_MATCH) INSERT_STRING(s, s->strstart, hash_head);
        s->prev_length = s->match_length; s->prev_match = s->match_start;
        s->match_length = MIN_MATCH - 1;
        if (hash_head != 0 && s->prev_length < (uint32_t)s->max_insert && s->strstart - hash_head <= MAX_DIST) {
            s->match_length = longest_match(s, hash_head);
            if (s->match_length <= 5 && s->match_length == MIN_MATCH && s->strstart - s->match_start > TOO_FAR)
                s->match_length = MIN_MATCH - 1;                /* a far 3-byte match costs more than 3 literals */
        }
        if (s->prev_length >= MIN_MATCH && s->match_length <= s->prev_length) {
            uint32_t max_ins = s->strstart + s->lookahead - MIN_MATCH;
            bflush = tally(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH);
            s->lookahead -= s->prev_length - 1;
            s->prev_length -= 2;
            do {
                if (++s->strstart <= max_ins) INSERT_STRING(s, s->strstart, hash_head);
            } while (--s->prev_length != 0);
            s->match_available = 0;
            s->match_length = MIN_MATCH - 1;
            s->strstart++;
            if (bflush) flush_block(s, 0);
        } else if (s->match_available) {
            bflush = tally(s, 0, s->window[s->strstart - 1]);
            if (bflush) flush_block(s, 0);                      /* the block ends before the byte still held */
            s->strstart++; s->lookahead--;
        } else {
            s->match_available = 1;
            s->strstart++; s->lookahead--;
        }
    }
    if (s->match_available) { tally(s, 0, s->window[s->strstart - 1]); s->match_available = 0; }
    if (final) { flush_block(s, 1); return; }
    if (s->last_lit) flush_block(s, 0);
    stored_block(s, NULL, 0, 0);
}

static void deflate_chunk(dstate_t *s, int final)
{
    if (s->lazy) deflate_slow_chunk(s, final); else deflate_fast_chunk(s, final);
}

static dstate_t *dstate_new(int level)
{
    /* zlib's configuration_table: good_length, max_lazy (= max_insert for the greedy levels), nice_length, max_chain */
    static const int cfg[10][4] = {{0, 0, 0, 0}, {4, 4, 8, 4}, {4, 5, 16, 8}, {4, 6, 32, 32}, {4, 4, 16, 16}, {8, 16, 32, 32},
                                   {8, 16, 128, 128}, {8, 32, 128, 256}, {32, 128, 258, 1024}, {32, 258, 258, 4096}};
    dstate_t *s;
    if (!tables_ready) tables_init();
    if (level < 1 || level > 9) return NULL;
    s = (dstate_t *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->good_match = cfg[level][0]; s->max_insert = cfg[level][1]; s->nice_match = cfg[level][2]; s->max_chain = cfg[level][3];
    s->lazy = level >= 4;
    s->prev_length = MIN_MATCH - 1;
    s->l_desc = (tdesc_t){s->dyn_ltree, static_ltree, extra_lbits, 257, L_CODES, 15, 0};
    s->d_desc = (tdesc_t){s->dyn_dtree, static_dtree, extra_dbits, 0, D_CODES, 15, 0};
    s->bl_desc = (tdesc_t){s->bl_tree, NULL, extra_blbits, 0, BL_CODES, 7, 0};
    s->match_length = MIN_MATCH - 1;
    init_block(s);
    return s;
}

size_t qzo_deflate_chunk(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, int level, int final)
{
    dstate_t *s = dstate_new(level); size_t r;
    if (!s) return (size_t)-1;
    s->in = src; s->avail_in = (uint32_t)n; s->out = dst; s->out_cap = cap;
    deflate_chunk(s, final);
    r = s->overflow ? (size_t)-1 : s->out_pos;
    free(s);
    return r;
}

size_t qzo_deflate_symbols(const uint8_t *src, size_t n, int level, uint8_t *lc, uint16_t *dist, size_t cap)
{
    dstate_t *s = dstate_new(level); size_t r; uint8_t *tmp;
    if (!s) return 0;
    tmp = (uint8_t *)malloc(n + n / 8 + 1024);
    s->in = src; s->avail_in = (uint32_t)n; s->out = tmp; s->out_cap = n + n / 8 + 1024;
    s->dump_lc = lc; s->dump_dist = dist; s->dump_cap = cap;
    deflate_chunk(s, 1);
    r = s->dump_n;
    free(tmp); free(s);
    return r;
}
