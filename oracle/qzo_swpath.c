/*
 * qzo_swpath.c — restatement of the QATzip software-fallback call sites
 * (framing + chunk loop) on top of the oracle codecs.  TEST INFRASTRUCTURE.
 *
 *   qzo_sw_compress    <- qzDeflateSWCompress   src/qatzip_sw.c:77-256
 *                         qzLZ4SWCompress        src/qatzip_sw.c:443-471
 *   qzo_sw_decompress  <- qzSWDecompressMulti*   src/qatzip_sw.c:394-441,539-577,659-695
 * Wire formats: SURVEY.md Appendix A (gzip header written by zlib itself:
 * XFL=4 at level 1, OS=3; gzip-ext header via deflateSetHeader with the
 * 'Q','Z' extra field and OS=255, src/qatzip_sw.c:61-75).
 *
 * Only single-shot streams are restated (the stream is opened and, when
 * last==1, closed by the same call; last==0 leaves it open-ended exactly like
 * the first call of a longer stream).
 */
#include "qzo.h"
#include <stdlib.h>
#include <string.h>

#define QZ_OK 0
#define QZ_FAIL (-2)
#define QZ_DATA_ERROR (-4)

static void wr16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr32(uint8_t *p, uint32_t v) { wr16(p, v); wr16(p + 2, v >> 16); }
static uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
static uint32_t rd32(const uint8_t *p) { return rd16(p) | rd16(p + 2) << 16; }

static unsigned hdr_size(int fmt)
{
    switch (fmt) {
    case QZO_DEFLATE_GZIP: return 10;
    case QZO_DEFLATE_GZIP_EXT: return 24;
    case QZO_DEFLATE_ZLIB: return 2;
    default: return 0;          /* RAW; 4B's prefix is handled apart */
    }
}

int qzo_sw_compress(int fmt, int level, uint32_t hw_buff_sz,
                    const uint8_t *src, uint32_t *src_len,
                    uint8_t *dst, uint32_t *dst_len, int last, unsigned long *crc)
{
    uint32_t left_in = *src_len, cap = *dst_len, total_in = 0, total_out = 0;
    uint8_t *base = dst;
    uint32_t run_crc = 0, run_adler = 1;   /* zlib's strm->adler */
    unsigned xfl = level == 9 ? 2 : (level < 2 ? 4 : 0);

    *src_len = 0; *dst_len = 0;
    if (fmt == QZO_LZ4_FH) {
        size_t r = qzo_lz4f_compress_frame(src, left_in, dst, cap);
        if (r == 0) return QZ_FAIL;
        *src_len = left_in; *dst_len = (uint32_t)r;
        return QZ_OK;
    }
    if (fmt == QZO_DEFLATE_4B) { if (cap < 4) return QZ_FAIL; dst += 4; cap -= 4; }

    /* header, emitted by the first deflate() call of the stream */
    if (cap < hdr_size(fmt)) return QZ_FAIL;
    if (fmt == QZO_DEFLATE_GZIP) {
        static const uint8_t h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
        memcpy(dst, h, 10); dst[8] = (uint8_t)xfl; total_out = 10;
    } else if (fmt == QZO_DEFLATE_GZIP_EXT) {
        static const uint8_t h[24] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0, 'Q', 'Z', 8, 0,
                                      0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(dst, h, 24); dst[8] = (uint8_t)xfl; total_out = 24;
    } else if (fmt == QZO_DEFLATE_ZLIB) {
        unsigned lf = level < 2 ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
        unsigned header = (8 + (7 << 4)) << 8 | lf << 6;
        header += 31 - header % 31;
        dst[0] = (uint8_t)(header >> 8); dst[1] = (uint8_t)header; total_out = 2;
    }

    do {
        uint32_t send = left_in > hw_buff_sz ? hw_buff_sz : left_in;
        int fin;
        size_t r;
        left_in -= send;
        fin = (left_in == 0 && last == 1);
        r = qzo_deflate_chunk(src + total_in, send, dst + total_out, cap - total_out, level, fin);
        if (r == (size_t)-1) return QZ_FAIL;
        total_out += (uint32_t)r;
        if (fmt == QZO_DEFLATE_GZIP || fmt == QZO_DEFLATE_GZIP_EXT)
            run_adler = run_crc = qzo_crc32(run_crc, src + total_in, send);
        else if (fmt == QZO_DEFLATE_ZLIB)
            run_adler = qzo_adler32(run_adler, src + total_in, send);
        total_in += send;
        if (fin) {                       /* trailer comes out of the same deflate(Z_FINISH) */
            if (fmt == QZO_DEFLATE_GZIP || fmt == QZO_DEFLATE_GZIP_EXT) {
                if (cap - total_out < 8) return QZ_FAIL;
                wr32(dst + total_out, run_crc); wr32(dst + total_out + 4, total_in); total_out += 8;
            } else if (fmt == QZO_DEFLATE_ZLIB) {
                if (cap - total_out < 4) return QZ_FAIL;
                dst[total_out] = (uint8_t)(run_adler >> 24); dst[total_out + 1] = (uint8_t)(run_adler >> 16);
                dst[total_out + 2] = (uint8_t)(run_adler >> 8); dst[total_out + 3] = (uint8_t)run_adler;
                total_out += 4;
            }
        }
        *src_len = total_in; *dst_len = total_out;
        /* crc out-param, src/qatzip_sw.c:217-230 (incl. its multi-chunk quirk) */
        if (crc) {
            if (fmt == QZO_DEFLATE_RAW) *crc = qzo_crc32((uint32_t)*crc, src + total_in - send, send);
            else if (*crc == 0) *crc = run_adler;
            else *crc = qzo_crc32_combine((uint32_t)*crc, run_adler, *src_len);
        }
    } while (left_in);

    if (last == 1) {
        if (fmt == QZO_DEFLATE_GZIP_EXT) {            /* src/qatzip_sw.c:238-241 */
            wr32(base + 16, total_in);
            wr32(base + 20, total_out - 24 - 8);
        } else if (fmt == QZO_DEFLATE_4B) {           /* src/qatzip_sw.c:242-244 */
            wr32(base, total_out);
            *dst_len += 4;
        }
    }
    return QZ_OK;
}

/* ---------------- decompress ---------------- */

/* one gzip / gzip-ext / zlib / raw / 4B member starting at src.
 * returns QZ_OK and consumed/produced, or an error */
static int one_deflate_member(int fmt, const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t cap,
                              uint32_t *used, uint32_t *made)
{
    uint32_t pos = 0; size_t iu = 0, ou = 0; int r;
    *used = *made = 0;
    if (fmt == QZO_DEFLATE_4B) { if (n < 4) return QZ_DATA_ERROR; pos = 4; }
    else if (fmt == QZO_DEFLATE_GZIP || fmt == QZO_DEFLATE_GZIP_EXT) {
        unsigned flg;
        if (n < 10 || src[0] != 0x1f || src[1] != 0x8b || src[2] != 8 || (src[3] & 0xe0)) return QZ_DATA_ERROR;
        flg = src[3]; pos = 10;
        if (flg & 4) { if (pos + 2 > n) return QZ_DATA_ERROR; pos += 2 + rd16(src + pos); }
        if (flg & 8) { while (pos < n && src[pos]) pos++; pos++; }
        if (flg & 16) { while (pos < n && src[pos]) pos++; pos++; }
        if (flg & 2) pos += 2;
        if (pos > n) return QZ_DATA_ERROR;
    } else if (fmt == QZO_DEFLATE_ZLIB) {
        if (n < 2 || (src[0] & 0x0f) != 8 || ((src[0] << 8 | src[1]) % 31) || (src[1] & 0x20)) return QZ_DATA_ERROR;
        pos = 2;
    }
    r = qzo_inflate_raw(src + pos, n - pos, dst, cap, &iu, &ou);
    if (r < 0) return QZ_DATA_ERROR;
    if (r > 0) return QZ_FAIL;      /* truncated input / short output: not restated */
    pos += (uint32_t)iu;
    if (fmt == QZO_DEFLATE_GZIP || fmt == QZO_DEFLATE_GZIP_EXT) {
        if (pos + 8 > n) return QZ_DATA_ERROR;
        if (rd32(src + pos) != qzo_crc32(0, dst, ou)) return QZ_DATA_ERROR;
        if (rd32(src + pos + 4) != (uint32_t)ou) return QZ_DATA_ERROR;
        pos += 8;
    } else if (fmt == QZO_DEFLATE_ZLIB) {
        uint32_t a;
        if (pos + 4 > n) return QZ_DATA_ERROR;
        a = (uint32_t)src[pos] << 24 | (uint32_t)src[pos + 1] << 16 | (uint32_t)src[pos + 2] << 8 | src[pos + 3];
        if (a != qzo_adler32(1, dst, ou)) return QZ_DATA_ERROR;
        pos += 4;
    }
    *used = pos; *made = (uint32_t)ou;
    return QZ_OK;
}

static int one_lz4_frame(const uint8_t *src, uint32_t n, uint8_t *dst, uint32_t cap, uint32_t *used, uint32_t *made)
{
    uint32_t pos, out = 0; unsigned flg, bd; int has_csize, has_ccheck, has_bcheck, has_dict;
    *used = *made = 0;
    if (n < 7 || rd32(src) != 0x184D2204u) return QZ_FAIL;
    flg = src[4]; bd = src[5]; (void)bd;
    if ((flg >> 6) != 1) return QZ_FAIL;
    has_bcheck = (flg >> 4) & 1; has_csize = (flg >> 3) & 1; has_ccheck = (flg >> 2) & 1; has_dict = flg & 1;
    pos = 6 + (has_csize ? 8 : 0) + (has_dict ? 4 : 0);
    if (pos + 1 > n) return QZ_FAIL;
    if (src[pos] != ((qzo_xxh32(src + 4, pos - 4, 0) >> 8) & 0xff)) return QZ_FAIL;
    pos++;
    for (;;) {
        uint32_t bh, bsz;
        if (pos + 4 > n) return QZ_FAIL;
        bh = rd32(src + pos); pos += 4;
        if (bh == 0) break;
        bsz = bh & 0x7fffffffu;
        if (pos + bsz > n) return QZ_FAIL;
        if (bh & 0x80000000u) {
            if (out + bsz > cap) return QZ_FAIL;
            memcpy(dst + out, src + pos, bsz); out += bsz;
        } else {
            /* linked blocks may reference earlier output: decode in place on the whole buffer */
            int r = qzo_lz4_decompress_block_prefix(src + pos, (int)bsz, dst + out, (int)(cap - out),
                                                    (flg & 0x20) ? 0 : out);
            if (r < 0) return QZ_FAIL;
            out += (uint32_t)r;
        }
        pos += bsz + (has_bcheck ? 4 : 0);
    }
    if (has_ccheck) {
        if (pos + 4 > n || rd32(src + pos) != qzo_xxh32(dst, out, 0)) return QZ_FAIL;
        pos += 4;
    }
    *used = pos; *made = out;
    return QZ_OK;
}

int qzo_sw_decompress(int fmt, const uint8_t *src, uint32_t *src_len, uint8_t *dst, uint32_t *dst_len)
{
    uint32_t n = *src_len, cap = *dst_len, ti = 0, to = 0; int ret = QZ_OK;
    *src_len = 0; *dst_len = 0;
    while (ti < n && to < cap) {
        uint32_t u, m;
        ret = fmt == QZO_LZ4_FH ? one_lz4_frame(src + ti, n - ti, dst + to, cap - to, &u, &m)
                                : one_deflate_member(fmt, src + ti, n - ti, dst + to, cap - to, &u, &m);
        if (ret != QZ_OK) { *src_len = 0; *dst_len = 0; return ret; }
        ti += u; to += m;
        *src_len = ti; *dst_len = to;
    }
    return ret;
}
