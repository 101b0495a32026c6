/*
 * qz_api.cpp — the qatzip.h application interface (include/qatzip.h) on top of the
 * device-resident MI355X backend (include/qzamd_device.h).
 *
 * Structure of the reference this stands in for (nothing is copied, the flow is
 * re-thought for a GPU that takes a whole call as one batch):
 *   request engine   src/qatzip.c:1874-2097 (compress), :2446-2671 (decompress)
 *   SW chunk loop    src/qatzip_sw.c:77-256 (one stream per call, header/trailer, CRC out-param)
 *   member loop      src/qatzip_sw.c:394-441 (decompress member after member)
 *   defaults/params  src/qatzip.c:2780-2928, src/qatzip_utils.c:395-886
 *   pinned memory    src/qatzip_mem.c:102-241
 * Where the QAT engine keeps <=32 chunks in flight on one instance (doCompressIn/Out,
 * src/qatzip.c:1483-1764), a call here is: one H2D copy, one batch of kernels over all
 * chunks, one D2H copy.  There is NO software fallback: without a GPU qzInit reports
 * QZ_NOSW_NO_HW and every request fails (the CPU oracle under oracle/ is test-only).
 *
 * Wire formats produced are those of the reference's SOFTWARE path:
 * one gzip / gzip-ext / zlib / 4B member per stream with a Z_FULL_FLUSH marker after
 * every hw_buff_sz chunk.
 */
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <map>
#include <set>
#include <vector>

#include "../../include/qatzip.h"
#include "../../include/qzamd_device.h"

enum { F_4B = 0, F_GZIP, F_GZIP_EXT, F_RAW, F_LZ4, F_LZ4S, F_ZLIB };   /* DataFormatInternal_T order, src/qatzip_internal.h:238-253 */

struct Params {
    QzHuffmanHdr_T huffman_hdr; QzDirection_T direction; int fmt; unsigned comp_lvl; unsigned char comp_algorithm;
    unsigned max_forks; unsigned char sw_backup; unsigned hw_buff_sz, strm_buff_sz, input_sz_thrshold,
    req_cnt_thrshold, wait_cnt_thrshold; QzPollingMode_T polling_mode; unsigned is_sensitive_mode;
    unsigned char stop_at_stream_end, zlib_format; qzLZ4SCallbackFn cb; void *cb_ext; unsigned lz4s_mini_match;
};

struct Sess {
    Params p;
    qzd_ctx *ctx;
    uint8_t *d_in, *d_out; size_t in_cap, out_cap;
    /* open deflate stream (qzCompress with last == 0 keeps it open, src/qatzip_sw.c:115,233-253) */
    bool open; uint32_t run_sum; uint64_t st_in, st_out;
    unsigned char end_of_stream;
    std::vector<uint32_t> lens, crcs, adlers;
    /* a member whose output did not fit the caller's destination: decoded once into d_hold, handed out over as many
     * calls as it takes (decompress_deflate) */
    uint8_t *d_hold; uint64_t hold_len, hold_pos;
    /* wire framing of the compress side: false = the software path's (one member per stream, a flush marker after every
     * chunk), true = the hardware path's (one complete member per hw_buff_sz chunk, src/qatzip.c:1691-1718) */
    bool hw_framing;
    std::vector<unsigned char> hw_stage;
};

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER, g_mem_lock = PTHREAD_MUTEX_INITIALIZER;
static int g_inited, g_ndev, g_next_dev;
static QzLogLevel_T g_log = LOG_WARNING;
static std::map<uintptr_t, size_t> g_pinned;     /* pinned allocations: base -> bytes */
static const Params k_factory = {QZ_HUFF_HDR_DEFAULT, QZ_DIRECTION_DEFAULT, F_GZIP_EXT, QZ_COMP_LEVEL_DEFAULT,
                                 QZ_COMP_ALGOL_DEFAULT, QZ_MAX_FORK_DEFAULT, QZ_SW_BACKUP_DEFAULT, QZ_HW_BUFF_SZ,
                                 QZ_STRM_BUFF_SZ_DEFAULT, QZ_COMP_THRESHOLD_DEFAULT, QZ_REQ_THRESHOLD_DEFAULT,
                                 QZ_WAIT_CNT_THRESHOLD_DEFAULT, QZ_PERIODICAL_POLLING, 0, 0, 0, NULL, NULL, 3};
static Params g_def = k_factory;

static void logmsg(QzLogLevel_T lvl, const char *fmt, ...)
{
    if (lvl > g_log) return;
    va_list ap; va_start(ap, fmt);
    fputs(lvl <= LOG_ERROR ? "[qatzip-amd error] " : "[qatzip-amd] ", stderr);
    vfprintf(stderr, fmt, ap); va_end(ap);
}

extern "C" QzLogLevel_T qzSetLogLevel(QzLogLevel_T level)
{
    QzLogLevel_T old = g_log;
    if ((int)level >= LOG_NONE && (int)level <= LOG_DEBUG3) g_log = level;
    return old;
}

/* ------------------------------------------------------------------ parameters */
static int check_common(const Params &p, bool lz4)
{
    if (p.direction > QZ_DIR_BOTH || p.sw_backup > 1) return QZ_PARAMS;
    if (p.hw_buff_sz < QZ_HW_BUFF_MIN_SZ || p.hw_buff_sz > QZ_HW_BUFF_MAX_SZ || (p.hw_buff_sz & (p.hw_buff_sz - 1))) return QZ_PARAMS;
    if (p.strm_buff_sz < QZ_STRM_BUFF_MIN_SZ || p.strm_buff_sz > QZ_STRM_BUFF_MAX_SZ) return QZ_PARAMS;
    if (p.input_sz_thrshold < QZ_COMP_THRESHOLD_MINIMUM) return QZ_PARAMS;
    if (p.req_cnt_thrshold < QZ_REQ_THRESHOLD_MINIMUM || p.req_cnt_thrshold > QZ_REQ_THRESHOLD_MAXIMUM) return QZ_PARAMS;
    if (p.comp_lvl < 1 || p.comp_lvl > (lz4 ? QZ_LZS_COMP_LVL_MAXIMUM : QZ_DEFLATE_COMP_LVL_MAXIMUM_Gen3)) return QZ_PARAMS;
    return QZ_OK;
}

static void from_common(Params &p, const QzSessionParamsCommon_T &c)
{
    p.direction = c.direction; p.comp_lvl = c.comp_lvl; p.comp_algorithm = c.comp_algorithm; p.max_forks = c.max_forks;
    p.sw_backup = c.sw_backup; p.hw_buff_sz = c.hw_buff_sz; p.strm_buff_sz = c.strm_buff_sz;
    p.input_sz_thrshold = c.input_sz_thrshold; p.req_cnt_thrshold = c.req_cnt_thrshold;
    p.wait_cnt_thrshold = c.wait_cnt_thrshold; p.polling_mode = c.polling_mode; p.is_sensitive_mode = c.is_sensitive_mode;
}
static void to_common(const Params &p, QzSessionParamsCommon_T &c)
{
    c.direction = p.direction; c.comp_lvl = p.comp_lvl; c.comp_algorithm = p.comp_algorithm; c.max_forks = p.max_forks;
    c.sw_backup = p.sw_backup; c.hw_buff_sz = p.hw_buff_sz; c.strm_buff_sz = p.strm_buff_sz;
    c.input_sz_thrshold = p.input_sz_thrshold; c.req_cnt_thrshold = p.req_cnt_thrshold;
    c.wait_cnt_thrshold = p.wait_cnt_thrshold; c.polling_mode = p.polling_mode; c.is_sensitive_mode = p.is_sensitive_mode;
}
static int pub_fmt(int f) { return f == F_4B ? QZ_DEFLATE_4B : f == F_GZIP ? QZ_DEFLATE_GZIP : f == F_RAW ? QZ_DEFLATE_RAW : QZ_DEFLATE_GZIP_EXT; }

static int legacy_to(Params &p, const QzSessionParams_T *s)
{
    if (s->huffman_hdr > QZ_STATIC_HDR || s->direction > QZ_DIR_BOTH || s->comp_algorithm != QZ_DEFLATE) return QZ_PARAMS;
    if (s->comp_lvl < QZ_DEFLATE_COMP_LVL_MINIMUM || s->comp_lvl > QZ_DEFLATE_COMP_LVL_MAXIMUM) return QZ_PARAMS;
    if ((unsigned)s->data_fmt >= QZ_FMT_NUM) return QZ_PARAMS;
    p = g_def;
    p.huffman_hdr = s->huffman_hdr; p.direction = s->direction; p.fmt = (int)s->data_fmt; p.comp_lvl = s->comp_lvl;
    p.comp_algorithm = s->comp_algorithm; p.max_forks = s->max_forks; p.sw_backup = s->sw_backup;
    p.hw_buff_sz = s->hw_buff_sz; p.strm_buff_sz = s->strm_buff_sz; p.input_sz_thrshold = s->input_sz_thrshold;
    p.req_cnt_thrshold = s->req_cnt_thrshold; p.wait_cnt_thrshold = s->wait_cnt_thrshold;
    p.zlib_format = 0; p.stop_at_stream_end = 0;
    return check_common(p, false);
}
static int deflate_to(Params &p, const QzSessionParamsDeflate_T *s)
{
    if (s->huffman_hdr > QZ_STATIC_HDR || (unsigned)s->data_fmt >= QZ_FMT_NUM || s->common_params.comp_algorithm != QZ_DEFLATE) return QZ_PARAMS;
    p = g_def;
    from_common(p, s->common_params);
    p.huffman_hdr = s->huffman_hdr; p.fmt = (int)s->data_fmt; p.zlib_format = 0; p.stop_at_stream_end = 0;
    return check_common(p, false);
}
static int lz4_to(Params &p, const QzSessionParamsLZ4_T *s)
{
    if (s->common_params.comp_algorithm != QZ_LZ4) return QZ_PARAMS;
    p = g_def;
    from_common(p, s->common_params);
    p.fmt = F_LZ4;
    return check_common(p, true);
}

extern "C" int qzGetDefaults(QzSessionParams_T *d)
{
    if (!d) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock);
    Params p = g_def;
    pthread_mutex_unlock(&g_lock);
    d->huffman_hdr = p.huffman_hdr; d->direction = p.direction;
    d->data_fmt = (QzDataFormat_T)pub_fmt(p.fmt); d->comp_lvl = p.comp_lvl; d->comp_algorithm = QZ_DEFLATE;
    d->max_forks = p.max_forks; d->sw_backup = p.sw_backup; d->hw_buff_sz = p.hw_buff_sz; d->strm_buff_sz = p.strm_buff_sz;
    d->input_sz_thrshold = p.input_sz_thrshold; d->req_cnt_thrshold = p.req_cnt_thrshold; d->wait_cnt_thrshold = p.wait_cnt_thrshold;
    return QZ_OK;
}
extern "C" int qzGetDefaultsDeflate(QzSessionParamsDeflate_T *d)
{
    if (!d) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock); Params p = g_def; pthread_mutex_unlock(&g_lock);
    to_common(p, d->common_params); d->common_params.comp_algorithm = QZ_DEFLATE;
    d->huffman_hdr = p.huffman_hdr; d->data_fmt = (QzDataFormat_T)pub_fmt(p.fmt);
    return QZ_OK;
}
extern "C" int qzGetDefaultsDeflateExt(QzSessionParamsDeflateExt_T *d)
{
    if (!d) return QZ_PARAMS;
    qzGetDefaultsDeflate(&d->deflate_params);
    pthread_mutex_lock(&g_lock); d->stop_decompression_stream_end = g_def.stop_at_stream_end; d->zlib_format = g_def.zlib_format; pthread_mutex_unlock(&g_lock);
    return QZ_OK;
}
extern "C" int qzGetDefaultsLZ4(QzSessionParamsLZ4_T *d)
{
    if (!d) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock); Params p = g_def; pthread_mutex_unlock(&g_lock);
    to_common(p, d->common_params); d->common_params.comp_algorithm = QZ_LZ4;
    return QZ_OK;
}
extern "C" int qzGetDefaultsLZ4S(QzSessionParamsLZ4S_T *d)
{
    if (!d) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock); Params p = g_def; pthread_mutex_unlock(&g_lock);
    to_common(p, d->common_params); d->common_params.comp_algorithm = QZ_LZ4s;
    d->qzCallback = p.cb; d->qzCallback_external = p.cb_ext; d->lz4s_mini_match = p.lz4s_mini_match;
    return QZ_OK;
}
extern "C" int qzSetDefaults(QzSessionParams_T *d)
{
    Params p;
    if (!d || legacy_to(p, d) != QZ_OK) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock); g_def = p; pthread_mutex_unlock(&g_lock);
    return QZ_OK;
}
extern "C" int qzSetDefaultsDeflate(QzSessionParamsDeflate_T *d)
{
    Params p;
    if (!d || deflate_to(p, d) != QZ_OK) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock); g_def = p; pthread_mutex_unlock(&g_lock);
    return QZ_OK;
}
extern "C" int qzSetDefaultsDeflateExt(QzSessionParamsDeflateExt_T *d)
{
    Params p;
    if (!d || deflate_to(p, &d->deflate_params) != QZ_OK) return QZ_PARAMS;
    p.stop_at_stream_end = d->stop_decompression_stream_end; p.zlib_format = d->zlib_format;
    if (p.zlib_format) p.fmt = F_ZLIB;
    pthread_mutex_lock(&g_lock); g_def = p; pthread_mutex_unlock(&g_lock);
    return QZ_OK;
}
extern "C" int qzSetDefaultsLZ4(QzSessionParamsLZ4_T *d)
{
    Params p;
    if (!d || lz4_to(p, d) != QZ_OK) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock); g_def = p; pthread_mutex_unlock(&g_lock);
    return QZ_OK;
}
extern "C" int qzSetDefaultsLZ4S(QzSessionParamsLZ4S_T *d) { (void)d; return QZ_NOT_SUPPORTED; }

/* ------------------------------------------------------------------ init / sessions */
extern "C" int qzInit(QzSession_T *sess, unsigned char sw_backup)
{
    if (!sess || sw_backup > 1) return QZ_PARAMS;
    pthread_mutex_lock(&g_lock);
    if (g_inited) { pthread_mutex_unlock(&g_lock); return QZ_DUPLICATE; }
    g_ndev = qzd_device_count();
    if (g_ndev <= 0) {
        pthread_mutex_unlock(&g_lock);
        logmsg(LOG_ERROR, "no MI355X visible and this build has no software path\n");
        sess->hw_session_stat = QZ_NOSW_NO_HW;
        return QZ_NOSW_NO_HW;
    }
    const char *e = getenv("QATZIP_AMD_DEVICE");
    g_next_dev = e ? atoi(e) % g_ndev : 0;
    g_inited = 1;
    pthread_mutex_unlock(&g_lock);
    return QZ_OK;
}

static int make_session(QzSession_T *sess, const Params &p)
{
    if (sess->internal) return QZ_DUPLICATE;
    Sess *s = new (std::nothrow) Sess();
    if (!s) { sess->hw_session_stat = QZ_NOSW_LOW_MEM; return QZ_NOSW_LOW_MEM; }
    s->p = p; s->ctx = NULL; s->d_in = s->d_out = NULL; s->in_cap = s->out_cap = 0;
    s->d_hold = NULL; s->hold_len = s->hold_pos = 0;
    s->open = false; s->run_sum = 0; s->st_in = s->st_out = 0; s->end_of_stream = 0;
    {
        const char *e = getenv("QATZIP_AMD_HW_FRAMING");
        s->hw_framing = e && e[0] == '1';
    }
    sess->internal = s;
    sess->hw_session_stat = g_inited ? QZ_OK : QZ_NONE;
    sess->thd_sess_stat = QZ_OK;
    sess->total_in = sess->total_out = 0;
    return QZ_OK;
}

extern "C" int qzSetupSession(QzSession_T *sess, QzSessionParams_T *params)
{
    QzSessionParams_T tmp; Params p;
    if (!sess) return QZ_PARAMS;
    if (!params) { qzGetDefaults(&tmp); params = &tmp; }
    if (legacy_to(p, params) != QZ_OK) return QZ_PARAMS;
    return make_session(sess, p);
}
extern "C" int qzSetupSessionDeflate(QzSession_T *sess, QzSessionParamsDeflate_T *params)
{
    QzSessionParamsDeflate_T tmp; Params p;
    if (!sess) return QZ_PARAMS;
    if (!params) { qzGetDefaultsDeflate(&tmp); params = &tmp; }
    if (deflate_to(p, params) != QZ_OK) return QZ_PARAMS;
    return make_session(sess, p);
}
extern "C" int qzSetupSessionDeflateExt(QzSession_T *sess, QzSessionParamsDeflateExt_T *params)
{
    QzSessionParamsDeflateExt_T tmp; Params p;
    if (!sess) return QZ_PARAMS;
    if (!params) { qzGetDefaultsDeflateExt(&tmp); params = &tmp; }
    if (deflate_to(p, &params->deflate_params) != QZ_OK) return QZ_PARAMS;
    p.stop_at_stream_end = params->stop_decompression_stream_end; p.zlib_format = params->zlib_format;
    if (p.zlib_format) p.fmt = F_ZLIB;
    return make_session(sess, p);
}
extern "C" int qzSetupSessionLZ4(QzSession_T *sess, QzSessionParamsLZ4_T *params)
{
    QzSessionParamsLZ4_T tmp; Params p;
    if (!sess) return QZ_PARAMS;
    if (!params) { qzGetDefaultsLZ4(&tmp); params = &tmp; }
    if (lz4_to(p, params) != QZ_OK) return QZ_PARAMS;
    return make_session(sess, p);
}
extern "C" int qzSetupSessionLZ4S(QzSession_T *sess, QzSessionParamsLZ4S_T *params)
{
    (void)params;
    if (!sess) return QZ_PARAMS;
    return QZ_NOT_SUPPORTED;        /* LZ4s is a QAT-2.0 hardware format with no software path in the reference */
}

static void async_drain(QzSession_T *sess);   /* queued qzCompress2/qzDecompress2 requests of a session finish first */

extern "C" int qzTeardownSession(QzSession_T *sess)
{
    if (!sess) return QZ_PARAMS;
    async_drain(sess);
    Sess *s = (Sess *)sess->internal;
    if (s) {
        if (s->ctx) {
            if (s->d_in) qzd_dev_free(s->ctx, s->d_in);
            if (s->d_out) qzd_dev_free(s->ctx, s->d_out);
            if (s->d_hold) qzd_dev_free(s->ctx, s->d_hold);
            qzd_destroy(s->ctx);
        }
        delete s;
        sess->internal = NULL;
    }
    return QZ_OK;
}
static int ensure_ready(QzSession_T *sess, Sess **out);

/* not in qatzip.h (include/qzamd_device.h): which of the reference's two wire framings the compress side of a session
 * writes - 0 the software path's (default: what the parity claim is about), 1 the hardware path's per-chunk members */
extern "C" int qzamd_set_hw_framing(QzSession_T *sess, int on)
{
    Sess *s = NULL;
    if (!sess) return QZ_PARAMS;
    int rc = ensure_ready(sess, &s);
    if (rc < 0) return rc;
    if (s->open) return QZ_FAIL;                                    /* not in the middle of a stream */
    s->hw_framing = on != 0;
    return QZ_OK;
}

extern "C" int qzClose(QzSession_T *sess) { return sess ? QZ_OK : QZ_PARAMS; }

extern "C" int qzGetStatus(QzSession_T *sess, QzStatus_T *st)
{
    if (!sess || !st) return QZ_PARAMS;
    memset(st, 0, sizeof(*st));
    int n = qzd_device_count();
    st->qat_hw_count = (unsigned short)(n > 0 ? n : 0);
    st->qat_service_init = g_inited ? 1 : 0; st->qat_mem_drvr = 1; st->qat_instance_attach = sess->internal ? 1 : 0;
    st->hw_session_status = sess->hw_session_stat;
    st->algo_hw[QZ_DEFLATE] = 1; st->algo_hw[QZ_LZ4] = 1;
    return QZ_OK;
}
extern "C" int qzGetDeflateEndOfStream(QzSession_T *sess, unsigned char *eos)
{
    if (!sess || !eos || !sess->internal) return QZ_PARAMS;
    *eos = ((Sess *)sess->internal)->end_of_stream;
    return QZ_OK;
}

/* lazy init + setup, the scenarios of include/qatzip.h:122-148 */
static int ensure_ready(QzSession_T *sess, Sess **out)
{
    int rc = qzInit(sess, 1);
    if (rc < 0) return rc;
    if (!sess->internal || sess->hw_session_stat == QZ_NONE) {
        if (!sess->internal) {
            pthread_mutex_lock(&g_lock); Params p = g_def; pthread_mutex_unlock(&g_lock);
            rc = make_session(sess, p);
            if (rc < 0) return rc;
        }
        sess->hw_session_stat = QZ_OK;
    }
    Sess *s = (Sess *)sess->internal;
    if (!s->ctx) {
        pthread_mutex_lock(&g_lock);
        int dev = getenv("QATZIP_AMD_DEVICE") ? g_next_dev : (g_next_dev++ % g_ndev);
        pthread_mutex_unlock(&g_lock);
        if (qzd_create(dev, &s->ctx) != QZD_OK) { sess->hw_session_stat = QZ_NO_INST_ATTACH; return QZ_NOSW_NO_INST_ATTACH; }
    }
    *out = s;
    return QZ_OK;
}

static int reserve(Sess *s, size_t in_n, size_t out_n)
{
    if (in_n > s->in_cap) {
        if (s->d_in) qzd_dev_free(s->ctx, s->d_in);
        s->in_cap = 0;
        s->d_in = (uint8_t *)qzd_dev_alloc(s->ctx, in_n + 4096);
        if (!s->d_in) return QZ_NOSW_LOW_MEM;
        s->in_cap = in_n;
    }
    if (out_n > s->out_cap) {
        if (s->d_out) qzd_dev_free(s->ctx, s->d_out);
        s->out_cap = 0;
        s->d_out = (uint8_t *)qzd_dev_alloc(s->ctx, out_n + 4096);
        if (!s->d_out) return QZ_NOSW_LOW_MEM;
        s->out_cap = out_n;
    }
    return QZ_OK;
}

static void wr32(unsigned char *p, uint32_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); p[2] = (unsigned char)(v >> 16); p[3] = (unsigned char)(v >> 24); }
static uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint32_t rd16(const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

static unsigned hdr_len(int fmt) { return fmt == F_GZIP ? 10 : fmt == F_GZIP_EXT ? 24 : fmt == F_ZLIB ? 2 : fmt == F_4B ? 4 : 0; }
static unsigned ftr_len(int fmt) { return (fmt == F_GZIP || fmt == F_GZIP_EXT) ? 8 : fmt == F_ZLIB ? 4 : 0; }

extern "C" uint32_t qzd_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);   /* qzd_device.hip */
static void wr32be(unsigned char *p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }

/* member header with the size fields still zero (src/qatzip_sw.c:61-75,158-171 and zlib's own gzip / zlib headers) */
static void write_header(unsigned char *dest, int fmt, unsigned lvl)
{
    const unsigned char xfl = lvl == 9 ? 2 : lvl < 2 ? 4 : 0;       /* zlib's gzip XFL: 2 = best, 4 = fastest */
    if (fmt == F_GZIP) { static const unsigned char h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 3}; memcpy(dest, h, 10); dest[8] = xfl; }
    else if (fmt == F_GZIP_EXT) { static const unsigned char h[24] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 4, 255, 12, 0, 'Q', 'Z', 8, 0}; memcpy(dest, h, 24); dest[8] = xfl; }
    else if (fmt == F_4B) wr32(dest, 0);
    else if (fmt == F_ZLIB) {                                       /* CMF deflate/32K window; FLG = FLEVEL by level + FCHECK */
        static const unsigned char flg[4] = {0x01, 0x5e, 0x9c, 0xda};
        dest[0] = 0x78; dest[1] = flg[lvl < 2 ? 0 : lvl < 6 ? 1 : lvl == 6 ? 2 : 3];
    }
}

/* ------------------------------------------------------------------ compress */
static bool pinned_span(const void *p, size_t len);
static int compress_deflate(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len,
                            unsigned char *dest, unsigned int *dest_len, unsigned int last, unsigned long *crc)
{
    const int fmt = s->p.fmt;
    const uint32_t n = *src_len, cap = *dest_len, hw = s->p.hw_buff_sz;
    const unsigned lvl = s->p.comp_lvl;
    if (lvl < 1 || lvl > 9) { logmsg(LOG_ERROR, "comp_lvl %u: zlib has levels 1-9\n", lvl); return QZ_NOT_SUPPORTED; }
    const bool opening = !s->open;
    const unsigned hl = opening ? hdr_len(fmt) : 0;
    *src_len = 0; *dest_len = 0;
    if (cap < hl) return QZ_BUF_ERROR;
    const uint32_t nchunks = n ? (n + hw - 1) / hw : 1;
    const uint64_t worst = (uint64_t)n + (uint64_t)nchunks * (5ull * (hw / 32767 + 2) + 16) + 64;
    int rc = reserve(s, n, worst);
    if (rc) return rc;
    uint64_t produced = 0;
    s->crcs.resize(nchunks); s->lens.resize(nchunks);
    /* A destination in qzMalloc(PINNED_MEM) memory that holds the worst case: the gather kernels write the stream
     * straight into it - the compressed bytes of a batch cross PCIe while the next batch is parsed, instead of as one
     * copy after the last kernel (the device can address hipHostMalloc memory; 1 GiB call: 8 ms of a 52 ms call). */
    const unsigned fl_all = last ? ftr_len(fmt) : 0;
    const bool direct = n >= (8u << 20) && fmt != F_ZLIB && (uint64_t)hl + worst + fl_all <= cap && pinned_span(dest + hl, cap - hl);
    uint8_t *const dev_dst = direct ? (uint8_t *)dest + hl : s->d_out;
    const uint64_t dev_cap = direct ? (uint64_t)cap - hl - fl_all : s->out_cap;
    if (qzd_deflate_raw_from_host(s->ctx, src, s->d_in, n, hw, (int)s->p.comp_lvl, (int)last, dev_dst, dev_cap, &produced, s->crcs.data()) != QZD_OK) {
        logmsg(LOG_ERROR, "GPU deflate failed: %s\n", qzd_last_error(s->ctx));
        return QZ_FAIL;
    }
    if (fmt == F_ZLIB) {                                            /* the zlib wrapper's trailer is an Adler-32, not a CRC */
        s->adlers.resize(nchunks);
        if (qzd_adler32_chunks(s->ctx, s->d_in, n, hw, s->adlers.data()) != QZD_OK) return QZ_FAIL;
    }
    /* how many whole chunks fit into dest (header now, trailer only with the final chunk)? */
    uint32_t take = nchunks; uint64_t bytes = produced;
    const unsigned fl = last ? ftr_len(fmt) : 0;
    if (hl + produced + fl > cap) {
        if (qzd_chunk_lens(s->ctx, s->lens.data(), nchunks) != QZD_OK) return QZ_FAIL;
        bytes = 0; take = 0;
        while (take < nchunks && hl + bytes + s->lens[take] <= cap) bytes += s->lens[take++];
        if (take == nchunks) take--, bytes -= s->lens[take];      /* the trailer is what does not fit */
        if (take == 0) return QZ_BUF_ERROR;
    }
    const bool complete = take == nchunks;
    const uint32_t used = complete ? n : take * hw;
    if (opening) {                                                  /* header, src/qatzip_sw.c:61-75,158-171 and zlib's own gzip header */
        write_header(dest, fmt, lvl);
        s->open = true; s->run_sum = fmt == F_ZLIB ? 1u : 0u; s->st_in = 0; s->st_out = hl;
    }
    if (bytes && !direct && qzd_d2h(s->ctx, dest + hl, s->d_out, bytes) != QZD_OK) return QZ_FAIL;
    /* running CRC-32 of the stream (zlib's strm->adler for gzip) + the crc out-parameter of
     * src/qatzip_sw.c:217-230, including its cumulative-fold behaviour on multi-chunk calls */
    uint32_t done_in = 0;
    for (uint32_t k = 0; k < take; k++) {
        uint32_t cl = std::min<uint32_t>(hw, n - k * hw);
        if (fmt == F_GZIP || fmt == F_GZIP_EXT) s->run_sum = qzd_crc32_combine(s->run_sum, s->crcs[k], cl);
        else if (fmt == F_ZLIB) s->run_sum = qzd_adler32_combine(s->run_sum, s->adlers[k], cl);
        done_in += cl;
        if (crc) {
            if (fmt == F_RAW) *crc = qzd_crc32_combine((uint32_t)*crc, s->crcs[k], cl);
            else {
                uint32_t adler = (fmt == F_4B) ? 1u : s->run_sum;
                if (*crc == 0) *crc = adler; else *crc = qzd_crc32_combine((uint32_t)*crc, adler, done_in);
            }
        }
    }
    s->st_in += used; s->st_out += bytes;
    uint32_t out_total = hl + (uint32_t)bytes;
    if (complete && last) {
        if (fmt == F_GZIP || fmt == F_GZIP_EXT) { wr32(dest + out_total, s->run_sum); wr32(dest + out_total + 4, (uint32_t)s->st_in); out_total += 8; s->st_out += 8; }
        if (fmt == F_ZLIB) { wr32be(dest + out_total, s->run_sum); out_total += 4; s->st_out += 4; }
        if (opening && fmt == F_GZIP_EXT) { wr32(dest + 16, (uint32_t)s->st_in); wr32(dest + 20, (uint32_t)(s->st_out - 24 - 8)); }
        if (opening && fmt == F_4B) wr32(dest, (uint32_t)(s->st_out - 4));
        s->open = false;
    }
    *src_len = used; *dest_len = out_total;
    sess->total_in += used; sess->total_out += out_total;
    return complete ? QZ_OK : QZ_BUF_ERROR;
}

/* The HARDWARE path's framing (qzamd_set_hw_framing / QATZIP_AMD_HW_FRAMING=1): every hw_buff_sz chunk becomes a
 * complete member of its own - header, a deflate stream that ends with BFINAL, footer - the way doCompressOut retires a
 * chunk (src/qatzip.c:1691-1718: outputHeaderGen, payload, outputFooterGen).  GZIP_EXT: qzGzipHeaderGen
 * (src/qatzip_gzip.c:98-118: XFL 0, OS 255, both sizes in the 'QZ' extra field) + CRC-32 / ISIZE of the chunk; GZIP:
 * stdGzipHeaderGen (:120-136) + the same footer; 4B: the chunk's compressed length (qz4BHeaderGen :138-143); RAW has no
 * framing and keeps the software path's stream.  `last` plays no part (every call is closed), the crc out-parameter is the
 * CRC-32 of the data (crc32_combine over the chunks, src/qatzip.c:1711).  Interop: this is what streams from QAT boxes
 * look like, and decompress_sized_members() reads thousands of such members in one launch. */
static int compress_deflate_hw(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len,
                               unsigned char *dest, unsigned int *dest_len, unsigned long *crc)
{
    const int fmt = s->p.fmt;
    const uint32_t n = *src_len, cap = *dest_len, hw = s->p.hw_buff_sz;
    *src_len = 0; *dest_len = 0;
    if (s->p.comp_lvl < 1 || s->p.comp_lvl > 9) return QZ_NOT_SUPPORTED;
    if (n == 0) return QZ_OK;                                       /* (not reached: calls below input_sz_thrshold take the software path) */
    const uint32_t nchunks = (n + hw - 1) / hw;
    const unsigned hl = fmt == F_GZIP_EXT ? 24 : fmt == F_GZIP ? 10 : 4, fl = fmt == F_4B ? 0 : 8;
    const uint64_t in_bytes = (uint64_t)nchunks * hw;
    const uint64_t worst = in_bytes + (uint64_t)nchunks * (5ull * (hw / 32767 + 2) + 16) + 64;
    int rc = reserve(s, in_bytes, worst);
    if (rc) return rc;
    if (qzd_h2d(s->ctx, s->d_in, src, n) != QZD_OK) return QZ_FAIL;
    std::vector<uint32_t> cdesc(nchunks), lens(nchunks), crcs(nchunks);
    for (uint32_t k = 0; k < nchunks; k++) cdesc[k] = std::min<uint32_t>(hw, n - k * hw) | 0x80000000u;    /* every chunk closes its stream */
    uint64_t produced = 0;
    if (qzd_deflate_slots(s->ctx, s->d_in, nchunks, hw, (int)s->p.comp_lvl, cdesc.data(), s->d_out, s->out_cap, &produced,
                          lens.data(), crcs.data()) != QZD_OK) {
        logmsg(LOG_ERROR, "GPU deflate failed: %s\n", qzd_last_error(s->ctx));
        return QZ_FAIL;
    }
    /* whole members that fit */
    uint32_t take = 0; uint64_t bytes = 0, body = 0;
    while (take < nchunks && bytes + hl + lens[take] + fl <= cap) { bytes += hl + lens[take] + fl; body += lens[take]; take++; }
    if (take == 0) return QZ_BUF_ERROR;
    try { s->hw_stage.resize(body); } catch (const std::bad_alloc &) { return QZ_NOSW_LOW_MEM; }
    if (body && qzd_d2h(s->ctx, s->hw_stage.data(), s->d_out, body) != QZD_OK) return QZ_FAIL;
    unsigned char *o = dest; const unsigned char *b = s->hw_stage.data();
    for (uint32_t k = 0; k < take; k++) {
        const uint32_t cl = cdesc[k] & 0x7fffffffu, zl = lens[k];
        if (fmt == F_GZIP_EXT) {
            static const unsigned char h[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 255, 12, 0, 'Q', 'Z', 8, 0};
            memcpy(o, h, 16); wr32(o + 16, cl); wr32(o + 20, zl);
        } else if (fmt == F_GZIP) {
            static const unsigned char h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 255};
            memcpy(o, h, 10);
        } else wr32(o, zl);
        memcpy(o + hl, b, zl);
        if (fl) { wr32(o + hl + zl, crcs[k]); wr32(o + hl + zl + 4, cl); }
        if (crc) *crc = (*crc == 0 && k == 0) ? crcs[k] : qzd_crc32_combine((uint32_t)*crc, crcs[k], cl);
        o += hl + zl + fl; b += zl;
    }
    const uint32_t used = take == nchunks ? n : take * hw;
    *src_len = used; *dest_len = (unsigned int)bytes;
    sess->total_in += used; sess->total_out += bytes;
    return take == nchunks ? QZ_OK : QZ_BUF_ERROR;
}

/* An LZ4 session in the HARDWARE path's framing (qzamd_set_hw_framing): every hw_buff_sz chunk is a frame of its own
 * behind qzLZ4HeaderGen's header - FLG 0x4C (blocks not marked independent), BD 64 KB, content size = the bytes the chunk
 * consumed, header checksum - and in front of qzLZ4FooterGen's end mark + XXH32 of the chunk (src/qatzip_lz4.c:104-143;
 * retired chunk by chunk, src/qatzip.c:1691-1718).  A chunk of at most 64 KB is one block; a larger hw_buff_sz gives the
 * frame several 64 KB blocks that may reach back into each other, which is what the linked-block kernel writes (its
 * header is this very one).  Whole frames that fit, QZ_BUF_ERROR with progress otherwise (the hardware contract). */
static int compress_lz4_hw(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len,
                           unsigned char *dest, unsigned int *dest_len)
{
    const uint32_t n = *src_len, cap = *dest_len, hw = s->p.hw_buff_sz;
    *src_len = 0; *dest_len = 0;
    if (s->p.comp_lvl >= 3) return QZ_NOT_SUPPORTED;
    const uint32_t nchunks = (n + hw - 1) / hw;
    const uint64_t per = 15 + 4ull * ((hw + 65535) >> 16) + hw + 8;       /* header, block headers, stored blocks, footer */
    int rc = reserve(s, n, (uint64_t)nchunks * per + 64);
    if (rc) return rc;
    if (qzd_h2d(s->ctx, s->d_in, src, n) != QZD_OK) return QZ_FAIL;
    std::vector<uint32_t> lens(nchunks);
    uint64_t produced = 0;
    /* one launch for all chunks, whatever hw_buff_sz: one-block frames up to 64 KB, linked blocks above (a wave per chunk) */
    if (qzd_lz4_compress_frames_hw(s->ctx, s->d_in, n, hw, s->d_out, s->out_cap, &produced, lens.data()) != QZD_OK) return QZ_FAIL;
    uint32_t take = 0; uint64_t bytes = 0;
    while (take < nchunks && bytes + lens[take] <= cap) bytes += lens[take++];
    if (take == 0) return QZ_BUF_ERROR;
    if (qzd_d2h(s->ctx, dest, s->d_out, bytes) != QZD_OK) return QZ_FAIL;
    const uint32_t used = take == nchunks ? n : take * hw;
    *src_len = used; *dest_len = (unsigned int)bytes;
    sess->total_in += used; sess->total_out += bytes;
    return take == nchunks ? QZ_OK : QZ_BUF_ERROR;
}

/* LZ4 sessions: one frame per call, `last` ignored (src/qatzip_sw.c:443-471) */
static int compress_lz4(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len,
                        unsigned char *dest, unsigned int *dest_len)
{
    const uint32_t n = *src_len, cap = *dest_len;
    *src_len = 0; *dest_len = 0;
    /* Up to 64 KB a call is one frame with one block; above, ONE frame whose 64 KB blocks are linked (each may reach
     * 64 KB back into the ones before it, one match table for the frame) - byte for byte what LZ4F_compressFrame writes
     * either way.  The linked frame is one serial chain, so it is one wave's work (qzd_lz4_compress_linked); callers that
     * want throughput make calls of at most 64 KB, like the reference's own harness (test/main.c:2204-2231) and its
     * hardware path, whose chunks are frames of their own (src/qatzip_lz4.c:104-132).  Only past 0x7fff0000 bytes, where
     * liblz4 starts rescaling its 32-bit positions, the call is written as one independent frame per 64 KB instead. */
    if (s->p.comp_lvl >= 3) return QZ_NOT_SUPPORTED;               /* level >= 3 is LZ4-HC in liblz4 */
    const bool linked = n > 65536 && n <= 0x7fff0000u;
    const uint64_t nfr = n ? ((uint64_t)n + 65535) >> 16 : 1;
    const uint64_t bound = linked ? 19 + 4 * nfr + (uint64_t)n + 8 : nfr * (19 + 4 + 8) + (uint64_t)n;   /* LZ4F_compressFrameBound */
    if (cap < bound) return QZ_FAIL;                               /* LZ4F_ERROR_dstMaxSize_tooSmall => QZ_FAIL */
    int rc = reserve(s, n, bound + 64);
    if (rc) return rc;
    if (n && qzd_h2d(s->ctx, s->d_in, src, n) != QZD_OK) return QZ_FAIL;
    uint64_t produced = 0;
    if (linked) { if (qzd_lz4_compress_linked(s->ctx, s->d_in, n, s->d_out, s->out_cap, &produced) != QZD_OK) return QZ_FAIL; }
    else if (qzd_lz4_compress_frames(s->ctx, s->d_in, n, 65536, s->d_out, s->out_cap, &produced, NULL) != QZD_OK) return QZ_FAIL;
    if (qzd_d2h(s->ctx, dest, s->d_out, produced) != QZD_OK) return QZ_FAIL;
    *src_len = n; *dest_len = (unsigned int)produced;
    sess->total_in += n; sess->total_out += produced;
    return QZ_OK;
}

/* walk one LZ4 frame on the host: total frame length and (if present) content size; <0 on malformed input */
static int64_t lz4_frame_extent(const unsigned char *p, uint32_t n, uint64_t *content, bool *has_content)
{
    if (n < 7 || rd32(p) != 0x184D2204u) return -1;
    unsigned flg = p[4];
    if ((flg >> 6) != 1) return -1;
    uint32_t pos = 6;
    *has_content = (flg >> 3) & 1; *content = 0;
    if (*has_content) { if (n < 14) return -1; *content = (uint64_t)rd32(p + 6) | (uint64_t)rd32(p + 10) << 32; pos += 8; }
    if (flg & 1) pos += 4;
    pos += 1;
    for (;;) {
        if (pos + 4 > n) return -1;
        uint32_t bh = rd32(p + pos); pos += 4;
        if (bh == 0) break;
        uint32_t bsz = bh & 0x7fffffffu;
        if (bsz > n - pos) return -1;
        pos += bsz + ((flg >> 4) & 1 ? 4 : 0);
    }
    if ((flg >> 2) & 1) { if (pos + 4 > n) return -1; pos += 4; }
    return pos;
}

static int decompress_lz4(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len,
                          unsigned char *dest, unsigned int *dest_len)
{
    const uint32_t n = *src_len, cap = *dest_len;
    *src_len = 0; *dest_len = 0;
    std::vector<qzd_lz4seg> segs;
    uint32_t ti = 0; uint64_t to = 0;
    while (ti < n && to < cap) {                                    /* frame loop, src/qatzip_sw.c:555-571 */
        uint64_t content; bool hc;
        int64_t ext = lz4_frame_extent(src + ti, n - ti, &content, &hc);
        if (ext < 0) {                                              /* SW path: LZ4F error => QZ_FAIL */
            if (segs.empty()) return QZ_FAIL;
            break;                                                  /* complete frames first; the caller comes back with the rest */
        }
        if (!hc) content = cap - to;                                /* unknown: give it the rest */
        if (to + content > cap) { if (segs.empty()) return QZ_BUF_ERROR; break; }
        qzd_lz4seg g; g.in_off = ti; g.out_off = to; g.in_len = (uint32_t)ext; g.out_cap = (uint32_t)content;
        segs.push_back(g);
        ti += (uint32_t)ext; to += content;
        if (!hc) break;                                             /* sizes unknown beyond this frame: one at a time */
    }
    if (segs.empty()) return QZ_OK;
    int rc = reserve(s, n, cap);
    if (rc) return rc;
    if (qzd_h2d(s->ctx, s->d_in, src, ti) != QZD_OK) return QZ_FAIL;
    std::vector<qzd_lz4res> res(segs.size());
    if (qzd_lz4_decompress_frames(s->ctx, s->d_in, s->d_out, segs.data(), (uint32_t)segs.size(), res.data()) != QZD_OK) return QZ_FAIL;
    uint64_t produced = 0;
    for (size_t i = 0; i < segs.size(); i++) {
        if (res[i].status != 0 || res[i].in_used != segs[i].in_len) return QZ_FAIL;
        if (i + 1 < segs.size() && res[i].out_len != segs[i].out_cap) return QZ_FAIL;
        produced = segs[i].out_off + res[i].out_len;
    }
    if (produced && qzd_d2h(s->ctx, dest, s->d_out, produced) != QZD_OK) return QZ_FAIL;
    *src_len = ti; *dest_len = (unsigned int)produced;
    sess->total_in += ti; sess->total_out += produced;
    return (ti < n && to >= cap) ? QZ_OK : QZ_OK;
}

/* Small synchronous calls from many threads share launches too (below: sync_via_queue) */
static bool sync_via_queue(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                           unsigned int *dest_len, unsigned long *crc, bool compress, int *rc_out);

/* the call itself, never through the queue (the queue's own fallback comes here) */
static int compress_direct(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                           unsigned int *dest_len, unsigned int last, unsigned long *crc, uint64_t *ext_rc, bool may_queue)
{
    int rc; Sess *s = NULL;
    if (!sess || !src || !src_len || !dest || !dest_len || (last != 0 && last != 1)) { rc = QZ_PARAMS; goto fail; }
    if (ext_rc) *ext_rc = 0;
    rc = ensure_ready(sess, &s);
    if (rc < 0) goto fail;
    if (may_queue && last == 1 && sync_via_queue(sess, s, src, src_len, dest, dest_len, crc, true, &rc)) {
        if (rc == QZ_OK || rc == QZ_BUF_ERROR) return rc;
        goto fail;
    }
    if (s->p.fmt == F_LZ4 && s->hw_framing && *src_len >= s->p.input_sz_thrshold) rc = compress_lz4_hw(sess, s, src, src_len, dest, dest_len);
    else if (s->p.fmt == F_LZ4) rc = compress_lz4(sess, s, src, src_len, dest, dest_len);
    else if (s->p.fmt == F_LZ4S) rc = QZ_UNSUPPORTED_FMT;
    /* the reference sends a call below input_sz_thrshold to its software path even on a QAT box (src/qatzip.c:1934-1947):
     * such a call - an empty one included - keeps the software path's framing */
    else if (s->hw_framing && !s->open && *src_len >= s->p.input_sz_thrshold &&
             (s->p.fmt == F_GZIP_EXT || s->p.fmt == F_GZIP || s->p.fmt == F_4B))
        rc = compress_deflate_hw(sess, s, src, src_len, dest, dest_len, crc);
    else rc = compress_deflate(sess, s, src, src_len, dest, dest_len, last, crc);
    sess->thd_sess_stat = rc;
    if (rc == QZ_OK || rc == QZ_BUF_ERROR) return rc;
fail:
    if (src_len) *src_len = 0;
    if (dest_len) *dest_len = 0;
    return rc;
}
extern "C" int qzCompressCrcExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                                unsigned int *dest_len, unsigned int last, unsigned long *crc, uint64_t *ext_rc)
{ return compress_direct(sess, src, src_len, dest, dest_len, last, crc, ext_rc, true); }
extern "C" int qzCompressExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                             unsigned int *dest_len, unsigned int last, uint64_t *ext_rc)
{
    if (!sess || (last != 0 && last != 1)) { if (src_len) *src_len = 0; if (dest_len) *dest_len = 0; return QZ_PARAMS; }
    return qzCompressCrcExt(sess, src, src_len, dest, dest_len, last, NULL, ext_rc);
}
extern "C" int qzCompress(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                          unsigned int *dest_len, unsigned int last)
{ return qzCompressExt(sess, src, src_len, dest, dest_len, last, NULL); }
extern "C" int qzCompressCrc(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                             unsigned int *dest_len, unsigned int last, unsigned long *crc)
{ return qzCompressCrcExt(sess, src, src_len, dest, dest_len, last, crc, NULL); }

/* ------------------------------------------------------------------ decompress */
/* parse one member header at p[0..n): returns payload offset or a negative QZ_* code;
 * for gzip-ext also the sizes its extra field carries (0 when absent) */
static int parse_header(int fmt, const unsigned char *p, uint32_t n, uint32_t *ext_src, uint32_t *ext_dst)
{
    *ext_src = *ext_dst = 0;
    if (fmt == F_RAW) return 0;
    if (fmt == F_4B) return n >= 4 ? 4 : QZ_DATA_ERROR;
    if (fmt == F_ZLIB) {
        if (n < 2 || (p[0] & 0x0f) != 8 || ((p[0] << 8 | p[1]) % 31) || (p[1] & 0x20)) return QZ_DATA_ERROR;
        return 2;
    }
    if (n < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return QZ_DATA_ERROR;
    uint32_t pos = 10; unsigned flg = p[3];
    if (flg & 4) {
        if (pos + 2 > n) return QZ_DATA_ERROR;
        uint32_t xl = rd16(p + pos);
        if (pos + 2 + xl > n) return QZ_DATA_ERROR;
        if (xl == 12 && p[pos + 2] == 'Q' && p[pos + 3] == 'Z' && rd16(p + pos + 4) == 8) { *ext_src = rd32(p + pos + 6); *ext_dst = rd32(p + pos + 10); }
        pos += 2 + xl;
    }
    if (flg & 8) { while (pos < n && p[pos]) pos++; pos++; }
    if (flg & 16) { while (pos < n && p[pos]) pos++; pos++; }
    if (flg & 2) pos += 2;
    return pos <= n ? (int)pos : QZ_DATA_ERROR;
}

/* Streams written by the reference's HARDWARE path (and by per-chunk callers of either path) are one gzip-ext member
 * per hw_buff_sz chunk, each header carrying both sizes (qzGzipHeaderGen, src/qatzip_gzip.c:86-118; the engine walks
 * them the same way in checkHeader, src/qatzip_utils.c:1232-1345).  The member boundaries are therefore known without
 * decoding anything: hop header to header on the host, then inflate every member as one segment of a single launch and
 * check all trailers with one CRC launch - the batch the QAT engine keeps in flight, made as wide as the call.
 * Returns the number of members done (0 = not this kind of stream, the member loop takes over), < 0 on error. */
static int decompress_sized_members(Sess *s, const unsigned char *src, uint32_t n, uint32_t cap, uint32_t *ti_out,
                                    uint32_t *to_out, unsigned long *crc, bool *resident)
{
    struct Mem { uint32_t pay, csz, usz; };
    std::vector<Mem> mem;
    uint32_t pos = 0; uint64_t out = 0;
    while (pos < n) {
        uint32_t es = 0, ed = 0;
        const int hl = parse_header(F_GZIP_EXT, src + pos, n - pos, &es, &ed);
        if (hl < 0 || es == 0 || ed == 0) break;                    /* not a sized member (e.g. a software-path stream) */
        if ((uint64_t)pos + hl + ed + 8 > n) break;                 /* cut short: the caller comes back with more */
        if (out + es > cap) break;                                  /* destination full: whole members only */
        if (es > 512u * 1024u) break;                               /* larger than any hw_buff_sz: a multi-chunk member of the software
                                                                     * path (qzCompress fills the sizes in for those too) - the member loop
                                                                     * below cuts it at its flush markers */
        Mem m = { pos + (uint32_t)hl, ed, es };
        mem.push_back(m);
        pos += (uint32_t)hl + ed + 8; out += es;
    }
    if (mem.size() < 2) return 0;
    const uint32_t nm = (uint32_t)mem.size();
    std::vector<qzd_infseg> segs(nm);
    std::vector<qzd_infres> res(nm);
    std::vector<qzd_range> rg(nm);
    std::vector<uint32_t> c32(nm);
    uint64_t oo = 0;
    for (uint32_t i = 0; i < nm; i++) {
        segs[i].in_off = mem[i].pay; segs[i].in_len = mem[i].csz; segs[i].out_off = oo; segs[i].out_cap = mem[i].usz;
        segs[i].flags = 0; segs[i].pad = mem[i].csz;               /* exact compressed length: lets phase A split it */
        rg[i].off = oo; rg[i].len = mem[i].usz; rg[i].pad = 0;
        oo += mem[i].usz;
    }
    if (!*resident) { *resident = true; if (qzd_h2d(s->ctx, s->d_in, src, n) != QZD_OK) return QZ_FAIL; }
    if (qzd_inflate_segments(s->ctx, s->d_in, s->d_out, segs.data(), nm, res.data()) != QZD_OK) return QZ_FAIL;
    if (qzd_crc32_ranges(s->ctx, s->d_out, rg.data(), nm, c32.data()) != QZD_OK) return QZ_FAIL;
    uint32_t good = 0; uint64_t to = 0;
    for (; good < nm; good++) {
        const Mem &m = mem[good];
        const unsigned char *tr = src + m.pay + m.csz;
        if (res[good].status != 0 || res[good].out_len != m.usz || res[good].in_used != m.csz) break;
        if (rd32(tr) != c32[good] || rd32(tr + 4) != m.usz) break;
        if (crc) *crc = (to == 0 && *crc == 0) ? c32[good] : qzd_crc32_combine((uint32_t)*crc, c32[good], m.usz);
        to += m.usz;
    }
    /* a member that is not ONE deflate segment (several chunks behind one sized header, or damaged) ends the batch; when it
     * is the first one the member loop takes over from the start and finds out which */
    if (good == 0) return 0;
    *ti_out = mem[good - 1].pay + mem[good - 1].csz + 8;            /* end of the last good member */
    *to_out = (uint32_t)to;
    return (int)good;
}

/* a member this large (compressed) is decoded piece by piece while it arrives: two pieces of qzd_inflate.hip's minimum */
#define QZ_PIPE_MIN_BYTES (24u << 20)
static int decompress_deflate(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len,
                              unsigned char *dest, unsigned int *dest_len, unsigned long *crc)
{
    const int fmt = s->p.fmt;
    const uint32_t n = *src_len, cap = *dest_len;
    if (s->d_hold) {
        /* the rest of a member that was larger than the destination (below): the next piece; the member's last input
         * byte was held back for exactly this, and goes with the last piece */
        const uint32_t k = (uint32_t)std::min<uint64_t>(cap, s->hold_len - s->hold_pos);
        if (k == 0 && s->hold_pos < s->hold_len) { *src_len = 0; *dest_len = 0; return QZ_BUF_ERROR; }
        if (k && qzd_d2h(s->ctx, dest, s->d_hold + s->hold_pos, k) != QZD_OK) return QZ_FAIL;
        if (crc && k) {
            uint32_t c = 0;
            if (qzd_crc32(s->ctx, s->d_hold + s->hold_pos, k, &c) != QZD_OK) return QZ_FAIL;
            *crc = qzd_crc32_combine((uint32_t)*crc, c, k);
        }
        s->hold_pos += k;
        const bool last_piece = s->hold_pos == s->hold_len && n >= 1;     /* the held-back byte has to be there to be consumed */
        if (last_piece) { qzd_dev_free(s->ctx, s->d_hold); s->d_hold = NULL; s->hold_len = s->hold_pos = 0; s->end_of_stream = 1; }
        *src_len = last_piece ? 1 : 0; *dest_len = k;
        sess->total_in += *src_len; sess->total_out += k;
        return QZ_OK;
    }
    int rc = reserve(s, n, cap);
    if (rc) return rc;
    /* the source goes to the device when something needs it there: a large member is decoded WHILE it arrives
     * (qzd_inflate_stream_from_host), everything else after one copy of the whole call */
    bool resident = false;
    uint32_t ti = 0, to = 0; int ret = QZ_OK;
    bool all_sent = true; uint64_t sent_bytes = 0;                  /* output the device layer has already put into dest */
    s->end_of_stream = 0;
    if (fmt == F_GZIP_EXT && !s->p.stop_at_stream_end) {
        const int done = decompress_sized_members(s, src, n, cap, &ti, &to, crc, &resident);
        if (done < 0) { *src_len = 0; *dest_len = 0; return done; }
        if (done > 0) s->end_of_stream = 1;
    }
    while (ti < n && to < cap) {                                    /* member loop, src/qatzip_sw.c:415-435 */
        uint32_t es, ed;
        int hl = parse_header(fmt, src + ti, n - ti, &es, &ed);
        if (hl < 0) { ret = hl; break; }
        uint64_t iu = 0, ol = 0; uint32_t c32 = 0;
        uint8_t *obuf = s->d_out + to;
        int sent = 0;                                               /* large members reach dest while they are still being decoded */
        int r;
        if (!resident && n - ti - hl >= QZ_PIPE_MIN_BYTES) {
            resident = true;                                        /* (the bytes before this member's payload are never read on the device) */
            r = qzd_inflate_stream_from_host(s->ctx, src + ti + hl, n - ti - hl, s->d_in + ti + hl, obuf, cap - to, s->p.hw_buff_sz, &iu, &ol,
                                             (fmt == F_GZIP || fmt == F_GZIP_EXT || crc) ? &c32 : NULL, dest + to, &sent);
        } else {
            if (!resident) { resident = true; if (qzd_h2d(s->ctx, s->d_in, src, n) != QZD_OK) { ret = QZ_FAIL; break; } }
            r = qzd_inflate_stream_to_host(s->ctx, s->d_in + ti + hl, n - ti - hl, obuf, cap - to, s->p.hw_buff_sz, &iu, &ol,
                                           (fmt == F_GZIP || fmt == F_GZIP_EXT || crc) ? &c32 : NULL, dest + to, &sent);
        }
        if (sent) sent_bytes += ol; else all_sent = false;
        bool held = false;
        if (r == QZD_ERR_DSTCAP && to == 0) {
            /* Not even the first member fits.  The software path would hand out what fits and keep its inflate state
             * (src/qatzip_sw.c:342-351, SURVEY 8b "decompress into 1 KB dest"); a segment-parallel decoder has no
             * such state to keep, so the member is decoded whole into a buffer of its own - size unknown, so grown
             * until it fits - and handed out piece by piece, this call and the following ones. */
            uint64_t hcap = std::max<uint64_t>(std::max<uint64_t>(2ull * cap, 8ull * (n - ti - hl)), 1u << 20);
            for (;;) {
                hcap = std::min<uint64_t>(hcap, 0xffffffffull);
                s->d_hold = (uint8_t *)qzd_dev_alloc(s->ctx, hcap + 4096);
                if (!s->d_hold) { r = QZD_ERR_DSTCAP; break; }
                r = qzd_inflate_stream(s->ctx, s->d_in + ti + hl, n - ti - hl, s->d_hold, hcap, s->p.hw_buff_sz, &iu, &ol,
                                       (fmt == F_GZIP || fmt == F_GZIP_EXT) ? &c32 : NULL);
                if (r == QZD_OK) { held = true; obuf = s->d_hold; break; }
                qzd_dev_free(s->ctx, s->d_hold); s->d_hold = NULL;
                if (r != QZD_ERR_DSTCAP || hcap >= 0xffffffffull) break;
                hcap *= 4;
            }
        }
        if (r == QZD_ERR_DSTCAP) { ret = QZ_BUF_ERROR; break; }
        /* only what the DATA is to blame for is a data error: a failed device call is QZ_FAIL, scratch that could not be
         * had QZ_NOSW_LOW_MEM (there is no software path behind this library to go to) */
        if (r == QZD_ERR_NOMEM) { ret = QZ_NOSW_LOW_MEM; break; }
        if (r == QZD_ERR_HIP || r == QZD_ERR_PARAM) { ret = QZ_FAIL; break; }
        if (r != QZD_OK) { ret = QZ_DATA_ERROR; break; }
        uint32_t pos = ti + (uint32_t)hl + (uint32_t)iu;
        if (fmt == F_GZIP || fmt == F_GZIP_EXT) {
            if (pos + 8 > n || rd32(src + pos) != c32 || rd32(src + pos + 4) != (uint32_t)ol) { ret = QZ_DATA_ERROR; break; }
            pos += 8;
        } else if (fmt == F_ZLIB) {
            const uint32_t R = 256 * 1024, nr = ol ? (uint32_t)((ol + R - 1) / R) : 1;
            std::vector<uint32_t> ad(nr);
            if (pos + 4 > n || qzd_adler32_chunks(s->ctx, obuf, ol, R, ad.data()) != QZD_OK) { ret = QZ_DATA_ERROR; break; }
            uint32_t a = 1;
            for (uint32_t k = 0; k < nr; k++) a = qzd_adler32_combine(a, ad[k], std::min<uint64_t>(R, ol - (uint64_t)k * R));
            const uint32_t want = (uint32_t)src[pos] << 24 | (uint32_t)src[pos + 1] << 16 | (uint32_t)src[pos + 2] << 8 | src[pos + 3];
            if (a != want) { ret = QZ_DATA_ERROR; break; }
            pos += 4;
        }
        if (held) {
            /* first piece now; everything but the member's last byte counts as consumed, so the caller keeps calling */
            const uint32_t k = (uint32_t)std::min<uint64_t>(cap, ol);
            if (k && qzd_d2h(s->ctx, dest, s->d_hold, k) != QZD_OK) { ret = QZ_FAIL; break; }
            if (crc && k) {
                uint32_t c = 0;
                if (qzd_crc32(s->ctx, s->d_hold, k, &c) != QZD_OK) { ret = QZ_FAIL; break; }
                *crc = *crc == 0 ? c : qzd_crc32_combine((uint32_t)*crc, c, k);
            }
            s->hold_len = ol; s->hold_pos = k;
            *src_len = pos - 1; *dest_len = k;
            sess->total_in += pos - 1; sess->total_out += k;
            return QZ_OK;
        }
        if (crc) *crc = (to == 0 && *crc == 0) ? c32 : qzd_crc32_combine((uint32_t)*crc, c32, ol);
        ti = pos; to += (uint32_t)ol;
        s->end_of_stream = 1;
        if (s->p.stop_at_stream_end) break;
    }
    if (ret != QZ_OK && s->d_hold) { qzd_dev_free(s->ctx, s->d_hold); s->d_hold = NULL; s->hold_len = s->hold_pos = 0; }
    /* a later member that is cut short or damaged does not undo the complete ones before it: partial consumption,
     * QZ_OK, and the next call (which starts at that member) reports the error - the member-granular version of what
     * the software path's kept inflate state does ("first half of the stream => QZ_OK, partial output") */
    if (ret == QZ_DATA_ERROR && ti > 0) ret = QZ_OK;
    if (ret != QZ_OK && !(ret == QZ_BUF_ERROR && to > 0)) { *src_len = 0; *dest_len = 0; return ret; }
    if (to && !(all_sent && sent_bytes == to) && qzd_d2h(s->ctx, dest, s->d_out, to) != QZD_OK) return QZ_FAIL;
    *src_len = ti; *dest_len = to;
    sess->total_in += ti; sess->total_out += to;
    return ret;
}

static int decompress_direct(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                             unsigned int *dest_len, unsigned long *crc, uint64_t *ext_rc, bool may_queue)
{
    int rc; Sess *s = NULL;
    if (!sess || !src || !src_len || !dest || !dest_len) { rc = QZ_PARAMS; goto fail; }
    if (ext_rc) *ext_rc = 0;
    if (*src_len == 0) { *dest_len = 0; return QZ_OK; }
    rc = ensure_ready(sess, &s);
    if (rc < 0) goto fail;
    if (may_queue && sync_via_queue(sess, s, src, src_len, dest, dest_len, crc, false, &rc)) {
        if (rc == QZ_OK || rc == QZ_BUF_ERROR) return rc;
        goto fail;
    }
    if (s->p.fmt == F_LZ4) rc = decompress_lz4(sess, s, src, src_len, dest, dest_len);
    else if (s->p.fmt == F_LZ4S) rc = QZ_UNSUPPORTED_FMT;
    else rc = decompress_deflate(sess, s, src, src_len, dest, dest_len, crc);
    sess->thd_sess_stat = rc;
    if (rc == QZ_OK || rc == QZ_BUF_ERROR) return rc;
fail:
    if (src_len) *src_len = 0;
    if (dest_len) *dest_len = 0;
    return rc;
}
extern "C" int qzDecompressCrcExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                                  unsigned int *dest_len, unsigned long *crc, uint64_t *ext_rc)
{ return decompress_direct(sess, src, src_len, dest, dest_len, crc, ext_rc, true); }
extern "C" int qzDecompress(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest, unsigned int *dest_len)
{ return qzDecompressCrcExt(sess, src, src_len, dest, dest_len, NULL, NULL); }
extern "C" int qzDecompressExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest, unsigned int *dest_len, uint64_t *ext_rc)
{ return qzDecompressCrcExt(sess, src, src_len, dest, dest_len, NULL, ext_rc); }
extern "C" int qzDecompressCrc(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest, unsigned int *dest_len, unsigned long *crc)
{ return qzDecompressCrcExt(sess, src, src_len, dest, dest_len, crc, NULL); }

/* src/qatzip.c:3022-3068: (9n/8 rounded up) + skid pad + one header/footer (the reference's `?:` precedence
 * makes its chunk count 0 or 1); LZ4: frame bound */
extern "C" unsigned int qzMaxCompressedLength(unsigned int src_sz, QzSession_T *sess)
{
    if (src_sz == 0) return QZ_COMPRESSED_SZ_OF_EMPTY_FILE;
    uint64_t out = ((uint64_t)9 * src_sz + 7) / 8 + QZ_SKID_PAD_SZ + (24 + 8);
    if (sess && sess->internal && ((Sess *)sess->internal)->p.fmt == F_LZ4)
        out = (uint64_t)src_sz + 27 + 4 * ((uint64_t)src_sz / 65536 + 1) + src_sz / 255;
    return (out >> 32) ? 0 : (unsigned int)out;
}

/* ------------------------------------------------------------------ pinned memory
 * src/qatzip_mem.c:102-241 over qaeMemAllocNUMA, here over hipHostMalloc: memory the GPU's DMA engines read and write
 * directly, so qzCompress / qzDecompress on such buffers copy asynchronously and overlap with the kernels. */
#include <sys/syscall.h>
#include <unistd.h>
#ifndef MPOL_PREFERRED
#define MPOL_DEFAULT 0
#define MPOL_PREFERRED 1
#endif
extern "C" void *qzd_host_alloc_pinned_numa(size_t n, int follow_policy);      /* qzd_device.hip */

extern "C" void *qzMalloc(size_t sz, int numa, int force_pinned)
{
    void *p = NULL;
    if (qzd_device_count() > 0) {
        /* numa >= 0: the pages come from that node (the reference passes the node to qaeMemAllocNUMA,
         * src/qatzip_mem.c:199-210) - the thread's memory policy is set for the duration of the allocation and
         * hipHostMalloc is told to follow it; numa < 0: the default local-node policy, which is what the reference's
         * "node of the current CPU" amounts to */
        bool policy = false;
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
        /* the calling thread's own policy (numactl --interleave / --membind ...) is put back afterwards, not MPOL_DEFAULT */
        int old_mode = MPOL_DEFAULT; unsigned long old_mask[16] = {0};
        if (numa >= 0 && numa < 1024 &&
            syscall(SYS_get_mempolicy, &old_mode, old_mask, (unsigned long)(sizeof(old_mask) * 8), NULL, 0ul) == 0) {
            unsigned long mask[16] = {0};
            mask[numa / (8 * sizeof(unsigned long))] = 1ul << (numa % (8 * sizeof(unsigned long)));
            policy = syscall(SYS_set_mempolicy, MPOL_PREFERRED, mask, (unsigned long)(sizeof(mask) * 8)) == 0;
        }
#endif
        p = qzd_host_alloc_pinned_numa(sz, policy ? 1 : 0);
#if defined(SYS_set_mempolicy) && defined(SYS_get_mempolicy)
        if (policy && syscall(SYS_set_mempolicy, old_mode, old_mode == MPOL_DEFAULT ? NULL : old_mask,
                              old_mode == MPOL_DEFAULT ? 0ul : (unsigned long)(sizeof(old_mask) * 8)) != 0)
            syscall(SYS_set_mempolicy, MPOL_DEFAULT, NULL, 0ul);
#endif
    }
    if (p) { pthread_mutex_lock(&g_mem_lock); g_pinned[(uintptr_t)p] = sz ? sz : 1; pthread_mutex_unlock(&g_mem_lock); return p; }
    return force_pinned == PINNED_MEM ? NULL : malloc(sz);
}
extern "C" void qzFree(void *m)
{
    if (!m) return;
    pthread_mutex_lock(&g_mem_lock);
    bool pinned = g_pinned.erase((uintptr_t)m) > 0;
    pthread_mutex_unlock(&g_mem_lock);
    if (pinned) qzd_host_free_pinned(m); else free(m);
}
/* 1 for ANY address inside a pinned allocation (the reference's page table marks every page of one,
 * src/qatzip_mem.c:102-149), 0 otherwise */
/* [p, p + len) lies inside one pinned allocation of qzMalloc */
static bool pinned_span(const void *p, size_t len)
{
    bool r = false;
    pthread_mutex_lock(&g_mem_lock);
    std::map<uintptr_t, size_t>::const_iterator it = g_pinned.upper_bound((uintptr_t)p);
    if (it != g_pinned.begin()) { --it; const uintptr_t o = (uintptr_t)p - it->first; r = o < it->second && len <= it->second - o; }
    pthread_mutex_unlock(&g_mem_lock);
    return r;
}

extern "C" int qzMemFindAddr(unsigned char *a)
{
    int r = 0;
    pthread_mutex_lock(&g_mem_lock);
    std::map<uintptr_t, size_t>::const_iterator it = g_pinned.upper_bound((uintptr_t)a);
    if (it != g_pinned.begin()) { --it; r = (uintptr_t)a - it->first < it->second ? 1 : 0; }
    pthread_mutex_unlock(&g_mem_lock);
    return r;
}

/* ------------------------------------------------------------------ streaming (src/qatzip_stream.c:403-781)
 * slab accumulate -> qzCompressCrc / qzDecompress on strm_buff_sz slabs -> drain */
struct StreamBuf { unsigned char *in, *out; unsigned cap, out_cap, out_off, in_off; bool flush_more; };

static int stream_init(Sess *s, QzStream_T *strm, bool comp)
{
    StreamBuf *b = (StreamBuf *)calloc(1, sizeof(StreamBuf));
    if (!b) return QZ_FAIL;
    b->cap = s->p.strm_buff_sz;
    b->out_cap = comp ? qzMaxCompressedLength(b->cap, NULL) + 64 : b->cap;
    b->in = (unsigned char *)qzMalloc(b->cap, 0, COMMON_MEM);
    b->out = (unsigned char *)qzMalloc(b->out_cap, 0, COMMON_MEM);
    if (!b->in || !b->out) { qzFree(b->in); qzFree(b->out); free(b); return QZ_FAIL; }
    strm->opaque = b; strm->pending_in = 0; strm->pending_out = 0; strm->crc_32 = 0;
    return QZ_OK;
}
static unsigned drain(QzStream_T *strm, StreamBuf *b, unsigned char *out, unsigned room)
{
    unsigned k = std::min(room, strm->pending_out);
    memcpy(out, b->out + b->out_off, k);
    b->out_off += k; strm->pending_out -= k;
    if (strm->pending_out == 0) b->out_off = 0;
    return k;
}

extern "C" int qzCompressStream(QzSession_T *sess, QzStream_T *strm, unsigned int last)
{
    if (!sess || !strm || (last != 0 && last != 1) || !strm->out || (!strm->in && strm->in_sz > 0)) {
        if (strm) { strm->in_sz = 0; strm->out_sz = 0; }
        return QZ_PARAMS;
    }
    Sess *s = NULL;
    if (ensure_ready(sess, &s) < 0) { strm->in_sz = 0; strm->out_sz = 0; return QZ_FAIL; }
    if (s->p.fmt != F_RAW && s->p.fmt != F_GZIP_EXT) { strm->in_sz = 0; strm->out_sz = 0; return QZ_PARAMS; }
    if (!strm->opaque && stream_init(s, strm, true) != QZ_OK) { strm->in_sz = 0; strm->out_sz = 0; return QZ_FAIL; }
    StreamBuf *b = (StreamBuf *)strm->opaque;
    unsigned consumed = 0, produced = 0; int rc = QZ_OK;
    const unsigned in_avail = strm->in_sz, out_room = strm->out_sz;
    for (;;) {
        if (strm->pending_out) {
            produced += drain(strm, b, strm->out + produced, out_room - produced);
            if (strm->pending_out) break;                           /* caller must bring more room */
        }
        unsigned k = std::min(in_avail - consumed, b->cap - strm->pending_in);
        if (k) { memcpy(b->in + strm->pending_in, strm->in + consumed, k); strm->pending_in += k; consumed += k; }
        const bool input_done = consumed == in_avail;
        if (strm->pending_in < b->cap && !(last && input_done)) break;   /* wait for a full slab */
        unsigned il = strm->pending_in, ol = b->out_cap;
        unsigned long c = strm->crc_32;
        const unsigned fin = (last && input_done) ? 1 : 0;
        rc = qzCompressCrc(sess, b->in, &il, b->out, &ol, fin, &c);
        if (rc != QZ_OK) { rc = QZ_FAIL; break; }
        strm->crc_32 = (unsigned int)c;
        strm->pending_in -= il; strm->pending_out = ol; b->out_off = 0;
        if (fin) { produced += drain(strm, b, strm->out + produced, out_room - produced); break; }
    }
    strm->in_sz = consumed; strm->out_sz = produced;
    return rc;
}

/* qzDecompressStream (src/qatzip_stream.c:596-781): the caller may hand over the compressed data in slices that cut
 * members anywhere, so unconsumed input is kept in the stream (pending_in) until the member it belongs to is complete;
 * the slab grows when a member is larger than strm_buff_sz.  Output goes straight to the caller's buffer - a member
 * larger than it comes out over several calls (decompress_deflate's held member), and pending_out says so. */
extern "C" int qzDecompressStream(QzSession_T *sess, QzStream_T *strm, unsigned int last)
{
    if (!sess || !strm || (last != 0 && last != 1) || !strm->out || (!strm->in && strm->in_sz > 0)) {
        if (strm) { strm->in_sz = 0; strm->out_sz = 0; }
        return QZ_PARAMS;
    }
    Sess *s = NULL;
    if (ensure_ready(sess, &s) < 0) { strm->in_sz = 0; strm->out_sz = 0; return QZ_FAIL; }
    if (s->p.fmt == F_LZ4S) { strm->in_sz = 0; strm->out_sz = 0; return QZ_PARAMS; }
    if (!strm->opaque && stream_init(s, strm, false) != QZ_OK) { strm->in_sz = 0; strm->out_sz = 0; return QZ_FAIL; }
    StreamBuf *b = (StreamBuf *)strm->opaque;
    const unsigned in_avail = strm->in_sz, out_room = strm->out_sz;
    /* take all of the caller's input into the slab */
    if ((uint64_t)strm->pending_in + in_avail > b->cap) {
        const uint64_t want = std::max<uint64_t>((uint64_t)strm->pending_in + in_avail, 2ull * b->cap);
        if (want > 0xfff00000ull) { strm->in_sz = 0; strm->out_sz = 0; return QZ_FAIL; }
        unsigned char *nb = (unsigned char *)qzMalloc((size_t)want, 0, COMMON_MEM);
        if (!nb) { strm->in_sz = 0; strm->out_sz = 0; return QZ_FAIL; }
        memcpy(nb, b->in, strm->pending_in);
        qzFree(b->in); b->in = nb; b->cap = (unsigned)want;
    }
    if (in_avail) memcpy(b->in + strm->pending_in, strm->in, in_avail);
    strm->pending_in += in_avail;
    unsigned produced = 0; int rc = QZ_OK;
    while (strm->pending_in > 0 && produced < out_room) {
        unsigned il = strm->pending_in, ol = out_room - produced;
        unsigned long c = strm->crc_32;
        rc = qzDecompressCrc(sess, b->in, &il, strm->out + produced, &ol, &c);
        if (rc == QZ_BUF_ERROR && (il > 0 || ol > 0)) rc = QZ_OK;
        if (rc != QZ_OK) break;
        strm->crc_32 = (unsigned int)c;
        produced += ol;
        if (il) { memmove(b->in, b->in + il, strm->pending_in - il); strm->pending_in -= il; }
        if (il == 0 && ol == 0) break;
    }
    strm->pending_out = s->d_hold ? (unsigned)std::min<uint64_t>(s->hold_len - s->hold_pos, 0xffffffffu) : 0;
    if (rc == QZ_BUF_ERROR) rc = QZ_OK;                             /* no room left: the caller comes back with an empty buffer */
    if ((rc == QZ_DATA_ERROR || (rc == QZ_FAIL && s->p.fmt == F_LZ4)) && !last) rc = QZ_OK;   /* an incomplete member / frame so far: wait for the rest of it */
    if (rc != QZ_OK) { strm->in_sz = 0; strm->out_sz = 0; return rc == QZ_DATA_ERROR ? QZ_DATA_ERROR : QZ_FAIL; }
    strm->in_sz = in_avail; strm->out_sz = produced;
    return QZ_OK;
}
extern "C" int qzEndStream(QzSession_T *sess, QzStream_T *strm)
{
    if (!sess || !strm) return QZ_PARAMS;
    StreamBuf *b = (StreamBuf *)strm->opaque;
    if (b) { qzFree(b->in); qzFree(b->out); free(b); strm->opaque = NULL; }
    strm->pending_in = 0; strm->pending_out = 0; strm->in_sz = 0; strm->out_sz = 0;
    return QZ_OK;
}

/* ------------------------------------------------------------------ qzCompress2 / qzDecompress2
 * src/qatzip.c:4112-4196: callback == NULL => the synchronous call (last = 1, optional input CRC-32 through
 * QzResult_T.crc); otherwise the request is queued and a library thread retires it and calls callback(res)
 * (the reference's ring + consumer thread, src/qatzip.c:3103-4110).  Here: one FIFO and one consumer thread per
 * process; requests retire in submission order, so a session's requests never run concurrently. */
struct AsyncReq { QzSession_T *sess; const unsigned char *src; unsigned char *dest; qzAsyncCallbackFn cb; QzResult_T *res; bool compress;
                  unsigned long *crcp;      /* a synchronous caller's crc in/out parameter (asynchronous ones carry it in res->crc) */
                  volatile int *done; };    /* a synchronous caller waits for this flag instead of a callback */
static pthread_cond_t g_aq_done = PTHREAD_COND_INITIALIZER;
static unsigned long *req_crc(const AsyncReq &q)
{
    if (q.crcp) return q.crcp;
    QzResult_T *r = q.res;
    return (r->crc && (r->crc->valid_flags & QZ_CRC32_VALID_MASK)) ? (unsigned long *)r->crc->in_crc.crc_32 : NULL;
}
static pthread_mutex_t g_aq_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_aq_more = PTHREAD_COND_INITIALIZER, g_aq_idle = PTHREAD_COND_INITIALIZER;
static std::vector<AsyncReq> g_aq;          /* pending, oldest first */
static size_t g_aq_head = 0;
static std::vector<QzSession_T *> g_aq_running;   /* sessions of the requests being executed (one launch can carry many) */
static bool g_aq_thread = false;

static int run_sync2(QzSession_T *sess, const unsigned char *src, unsigned char *dest, QzResult_T *r, bool compress, unsigned long *crcp = NULL)
{
    unsigned long *crc = crcp ? crcp : (r->crc && (r->crc->valid_flags & QZ_CRC32_VALID_MASK)) ? (unsigned long *)r->crc->in_crc.crc_32 : NULL;
    int rc = compress ? compress_direct(sess, src, &r->src_len, dest, &r->dest_len, 1, crc, &r->ext_rc, false)
                      : decompress_direct(sess, src, &r->src_len, dest, &r->dest_len, crc, &r->ext_rc, false);
    r->status = rc;
    return rc;
}

/* ---- the submission queue coalesces.  A 64 KB request alone keeps one wave of the GPU busy for milliseconds; what
 * the asynchronous API is for is many requests in flight, and whatever is waiting when the consumer comes round goes
 * to the device as ONE launch (qzd_deflate_slots): every request keeps its own member, header, trailer and CRC, and its
 * bytes are what a call of its own would have produced. */
#define AQ_BATCH_MAX_REQ   (4u << 20)      /* larger requests fill the device by themselves */
#define AQ_BATCH_MAX_SLOTS 16384u
#define AQ_BATCH_MAX_BYTES (1ull << 30)    /* slot-aligned input of one coalesced launch (a slot is hw_buff_sz bytes) */
static size_t g_aq_batches, g_aq_batched_reqs;       /* observability (qzamd_async_stats) */

/* pinned staging of the coalescing queue: one buffer per process, grown on demand, used only by the consumer thread */
static unsigned char *g_stage; static uint64_t g_stage_cap;
static unsigned char *stage_reserve(uint64_t n)
{
    if (n <= g_stage_cap) return g_stage;
    if (g_stage) qzd_host_free_pinned(g_stage);
    g_stage_cap = 0;
    n = (n + (1u << 20)) & ~(uint64_t)((1u << 20) - 1);
    g_stage = (unsigned char *)qzd_host_alloc_pinned((size_t)n);
    if (g_stage) g_stage_cap = n;
    return g_stage;
}

static Sess *batchable(const AsyncReq &q)
{
    Sess *s = NULL;
    if (!q.compress || !q.sess || !q.res || ensure_ready(q.sess, &s) < 0 || !s) return NULL;
    const int f = s->p.fmt;
    if (!(f == F_GZIP || f == F_GZIP_EXT || f == F_RAW || f == F_4B) || s->open) return NULL;
    /* a hardware-framing session writes one complete member per chunk (compress_deflate_hw); compress_batch writes the
     * software path's framing, so such a request runs alone - its framing must not depend on what else was queued */
    if (s->hw_framing) return NULL;
    if (s->p.comp_lvl < 1 || s->p.comp_lvl > 9 || q.res->src_len > AQ_BATCH_MAX_REQ) return NULL;
    return s;
}

/* run[0..n) are batchable with equal (fmt, level, hw_buff_sz).  Returns false if the batch could not be run as a
 * whole (nothing is reported then; the caller falls back to one call per request). */
static bool compress_batch(const std::vector<AsyncReq> &run, const std::vector<Sess *> &ss)
{
    Sess *s0 = ss[0];
    const int fmt = s0->p.fmt; const uint32_t hw = s0->p.hw_buff_sz; const unsigned lvl = s0->p.comp_lvl;
    std::vector<uint32_t> cdesc, first(run.size());
    for (size_t i = 0; i < run.size(); i++) {
        const uint32_t n = run[i].res->src_len, nch = n ? (n + hw - 1) / hw : 1;
        first[i] = (uint32_t)cdesc.size();
        for (uint32_t k = 0; k < nch; k++) cdesc.push_back(std::min<uint32_t>(hw, n - k * hw) | (k + 1 == nch ? 0x80000000u : 0));
    }
    const uint32_t nslots = (uint32_t)cdesc.size();
    const uint64_t in_bytes = (uint64_t)nslots * hw;
    const uint64_t worst = in_bytes + (uint64_t)nslots * (5ull * (hw / 32767 + 2) + 16) + 64;
    if (in_bytes > AQ_BATCH_MAX_BYTES + AQ_BATCH_MAX_REQ + hw || reserve(s0, in_bytes, worst) != QZ_OK) return false;
    /* slot-aligned copy of every request's input in the queue's pinned staging buffer (allocated once, reused, never
     * zeroed: the gaps between requests are not read - cdesc carries each slot's length), then ONE copy to the device */
    unsigned char *stage = stage_reserve(std::max<uint64_t>(in_bytes, worst));
    if (!stage) return false;
    for (size_t i = 0; i < run.size(); i++)
        if (run[i].res->src_len) memcpy(stage + (size_t)first[i] * hw, run[i].src, run[i].res->src_len);
    if (qzd_h2d(s0->ctx, s0->d_in, stage, in_bytes) != QZD_OK) return false;
    std::vector<uint32_t> lens(nslots), crcs(nslots);
    uint64_t produced = 0;
    if (qzd_deflate_slots(s0->ctx, s0->d_in, nslots, hw, (int)lvl, cdesc.data(), s0->d_out, s0->out_cap, &produced,
                          lens.data(), crcs.data()) != QZD_OK) {
        logmsg(LOG_ERROR, "coalesced GPU deflate failed: %s\n", qzd_last_error(s0->ctx));
        return false;
    }
    if (produced && qzd_d2h(s0->ctx, stage, s0->d_out, produced) != QZD_OK) return false;
    const unsigned hl = hdr_len(fmt), fl = ftr_len(fmt);
    uint64_t pos = 0;
    for (size_t i = 0; i < run.size(); i++) {
        QzResult_T *r = run[i].res; unsigned char *dest = run[i].dest;
        const uint32_t n = r->src_len, nch = n ? (n + hw - 1) / hw : 1, k0 = first[i];
        uint64_t body = 0;
        for (uint32_t k = 0; k < nch; k++) body += lens[k0 + k];
        const uint64_t bpos = pos; pos += body;
        if (hl + body + fl > r->dest_len) {                     /* too small for the whole member: the one-call path knows the partial-progress rules */
            run_sync2(run[i].sess, run[i].src, dest, r, true, run[i].crcp);
            continue;
        }
        unsigned long *crc = req_crc(run[i]);
        write_header(dest, fmt, lvl);
        memcpy(dest + hl, stage + bpos, body);
        uint32_t sum = 0, done = 0;
        for (uint32_t k = 0; k < nch; k++) {                    /* same folds as compress_deflate */
            const uint32_t cl = cdesc[k0 + k] & 0x7fffffffu;
            if (fmt == F_GZIP || fmt == F_GZIP_EXT) sum = qzd_crc32_combine(sum, crcs[k0 + k], cl);
            done += cl;
            if (crc) {
                if (fmt == F_RAW) *crc = qzd_crc32_combine((uint32_t)*crc, crcs[k0 + k], cl);
                else {
                    const uint32_t adler = fmt == F_4B ? 1u : sum;
                    if (*crc == 0) *crc = adler; else *crc = qzd_crc32_combine((uint32_t)*crc, adler, done);
                }
            }
        }
        uint32_t total = hl + (uint32_t)body;
        if (fmt == F_GZIP || fmt == F_GZIP_EXT) { wr32(dest + total, sum); wr32(dest + total + 4, n); total += 8; }
        if (fmt == F_GZIP_EXT) { wr32(dest + 16, n); wr32(dest + 20, (uint32_t)body); }
        if (fmt == F_4B) wr32(dest, (uint32_t)body);
        r->dest_len = total; r->ext_rc = 0; r->status = QZ_OK;  /* src_len: all of it */
        run[i].sess->total_in += n; run[i].sess->total_out += total; run[i].sess->thd_sess_stat = QZ_OK;
    }
    return true;
}

/* the other direction: a request whose source is exactly one gzip-ext member with its sizes in the header (what
 * qzCompress writes for a last = 1 call, and what the hardware path writes per chunk) can share a launch too - one
 * segment per request, all decoded at once */
static Sess *batchable_d(const AsyncReq &q, uint32_t *pay, uint32_t *csz, uint32_t *usz)
{
    Sess *s = NULL;
    if (q.compress || !q.sess || !q.res || ensure_ready(q.sess, &s) < 0 || !s) return NULL;
    if (!(s->p.fmt == F_GZIP_EXT || s->p.fmt == F_GZIP) || q.res->src_len > AQ_BATCH_MAX_REQ) return NULL;
    uint32_t es = 0, ed = 0;
    const int hl = parse_header(F_GZIP_EXT, q.src, q.res->src_len, &es, &ed);
    if (hl < 0 || es == 0 || ed == 0 || (uint64_t)hl + ed + 8 != q.res->src_len || es > q.res->dest_len || es > AQ_BATCH_MAX_REQ) return NULL;
    *pay = (uint32_t)hl; *csz = ed; *usz = es;
    return s;
}

static bool decompress_batch(const std::vector<AsyncReq> &run, const std::vector<Sess *> &ss)
{
    Sess *s0 = ss[0];
    const uint32_t nm = (uint32_t)run.size();
    std::vector<qzd_infseg> segs(nm);
    std::vector<qzd_infres> res(nm);
    std::vector<qzd_range> rg(nm);
    std::vector<uint32_t> c32(nm), pay(nm);
    uint64_t io = 0, oo = 0;
    for (uint32_t i = 0; i < nm; i++) {
        uint32_t csz = 0, usz = 0;
        if (!batchable_d(run[i], &pay[i], &csz, &usz)) return false;
        segs[i].in_off = io; segs[i].in_len = csz; segs[i].out_off = oo; segs[i].out_cap = usz; segs[i].flags = 0; segs[i].pad = csz;
        rg[i].off = oo; rg[i].len = usz; rg[i].pad = 0;
        io += (csz + 15u) & ~15u; oo += usz;
    }
    if (reserve(s0, io + 64, oo + 64) != QZ_OK) return false;
    unsigned char *stage = stage_reserve(std::max<uint64_t>(io + 64, oo));
    if (!stage) return false;
    for (uint32_t i = 0; i < nm; i++) memcpy(stage + segs[i].in_off, run[i].src + pay[i], segs[i].in_len);
    if (qzd_h2d(s0->ctx, s0->d_in, stage, io + 64) != QZD_OK) return false;
    if (qzd_inflate_segments(s0->ctx, s0->d_in, s0->d_out, segs.data(), nm, res.data()) != QZD_OK) return false;
    if (qzd_crc32_ranges(s0->ctx, s0->d_out, rg.data(), nm, c32.data()) != QZD_OK) return false;
    if (oo && qzd_d2h(s0->ctx, stage, s0->d_out, oo) != QZD_OK) return false;
    for (uint32_t i = 0; i < nm; i++) {
        QzResult_T *r = run[i].res;
        const unsigned char *tr = run[i].src + pay[i] + segs[i].in_len;
        const uint32_t usz = segs[i].out_cap;
        if (res[i].status != 0 || res[i].out_len != usz || res[i].in_used != segs[i].in_len || rd32(tr) != c32[i] || rd32(tr + 4) != usz) {
            run_sync2(run[i].sess, run[i].src, run[i].dest, r, false, run[i].crcp);      /* the one-call path reports what is wrong with it */
            continue;
        }
        memcpy(run[i].dest, stage + segs[i].out_off, usz);
        unsigned long *crc = req_crc(run[i]);
        if (crc) *crc = *crc == 0 ? c32[i] : qzd_crc32_combine((uint32_t)*crc, c32[i], usz);
        r->dest_len = usz; r->ext_rc = 0; r->status = QZ_OK;        /* src_len: the whole member */
        ss[i]->end_of_stream = 1;
        run[i].sess->total_in += r->src_len; run[i].sess->total_out += usz; run[i].sess->thd_sess_stat = QZ_OK;
    }
    return true;
}

extern "C" void qzamd_async_stats(uint64_t *launches, uint64_t *requests)
{
    pthread_mutex_lock(&g_aq_lock);
    if (launches) *launches = g_aq_batches;
    if (requests) *requests = g_aq_batched_reqs;
    pthread_mutex_unlock(&g_aq_lock);
}

static void *async_consumer(void *)
{
    std::vector<AsyncReq> run; std::vector<Sess *> ss;
    for (;;) {
        pthread_mutex_lock(&g_aq_lock);
        while (g_aq_head == g_aq.size()) {
            if (g_aq_head) { g_aq.clear(); g_aq_head = 0; }
            pthread_cond_broadcast(&g_aq_idle);
            pthread_cond_wait(&g_aq_more, &g_aq_lock);
        }
        /* everything that is waiting and can share a launch with the oldest request - in order, so requests still
         * retire in submission order */
        run.clear(); ss.clear();
        run.push_back(g_aq[g_aq_head++]);
        g_aq_running.assign(1, run[0].sess);
        pthread_mutex_unlock(&g_aq_lock);
        uint32_t a_ = 0, b_ = 0, c_ = 0;
        const bool comp = run[0].compress;
        Sess *s0 = comp ? batchable(run[0]) : batchable_d(run[0], &a_, &b_, &c_);
        if (s0) {
            ss.push_back(s0);
            uint64_t slots = run[0].res->src_len / s0->p.hw_buff_sz + 1;
            pthread_mutex_lock(&g_aq_lock);
            while (g_aq_head < g_aq.size() && slots < AQ_BATCH_MAX_SLOTS && slots * s0->p.hw_buff_sz < AQ_BATCH_MAX_BYTES) {
                AsyncReq q = g_aq[g_aq_head];
                pthread_mutex_unlock(&g_aq_lock);                /* ensure_ready may take the global lock */
                Sess *s = q.compress != comp ? NULL : comp ? batchable(q) : batchable_d(q, &a_, &b_, &c_);
                /* one launch runs on one GPU: only sessions of the oldest request's device ride along */
                const bool ok = s && qzd_ctx_device(s->ctx) == qzd_ctx_device(s0->ctx) &&
                                (!comp || (s->p.fmt == s0->p.fmt && s->p.comp_lvl == s0->p.comp_lvl && s->p.hw_buff_sz == s0->p.hw_buff_sz));
                pthread_mutex_lock(&g_aq_lock);
                if (!ok) break;
                run.push_back(q); ss.push_back(s); g_aq_head++;
                g_aq_running.push_back(q.sess);
                slots += q.res->src_len / s0->p.hw_buff_sz + 1;
            }
            pthread_mutex_unlock(&g_aq_lock);
        }
        bool done = false;
        if (run.size() > 1) {
            try { done = comp ? compress_batch(run, ss) : decompress_batch(run, ss); }
            catch (const std::bad_alloc &) { done = false; }        /* out of host memory for the batch: one request at a time */
            if (done) { pthread_mutex_lock(&g_aq_lock); g_aq_batches++; g_aq_batched_reqs += run.size(); pthread_mutex_unlock(&g_aq_lock); }
        }
        if (!done) for (size_t i = 0; i < run.size(); i++) run_sync2(run[i].sess, run[i].src, run[i].dest, run[i].res, run[i].compress, run[i].crcp);
        /* every request of the batch is finished before the first callback runs, and none of their sessions counts as
         * running any more: a callback may tear down its own session, or another one of the same batch */
        pthread_mutex_lock(&g_aq_lock);
        g_aq_running.clear();
        for (size_t i = 0; i < run.size(); i++) if (run[i].done) *run[i].done = 1;      /* synchronous callers of this batch */
        pthread_cond_broadcast(&g_aq_idle);
        pthread_cond_broadcast(&g_aq_done);
        pthread_mutex_unlock(&g_aq_lock);
        for (size_t i = 0; i < run.size(); i++) if (run[i].cb) run[i].cb(run[i].res);
    }
    return NULL;
}

/* wait until no queued or running request refers to sess (NULL: until the queue is empty) */
static pthread_t g_aq_tid;
static void async_drain(QzSession_T *sess)
{
    pthread_mutex_lock(&g_aq_lock);
    if (g_aq_thread && pthread_equal(pthread_self(), g_aq_tid)) {
        /* called from a completion callback (the consumer thread itself): waiting would wait for this very thread.
         * Requests of the session that are still queued are cancelled - reported QZ_FAIL through their callbacks. */
        std::vector<AsyncReq> dropped;
        size_t w = g_aq_head;
        for (size_t i = g_aq_head; i < g_aq.size(); i++) {
            if (sess && g_aq[i].sess != sess) g_aq[w++] = g_aq[i]; else dropped.push_back(g_aq[i]);
        }
        g_aq.resize(w);
        pthread_mutex_unlock(&g_aq_lock);
        for (size_t i = 0; i < dropped.size(); i++) {
            QzResult_T *r = dropped[i].res;
            r->status = QZ_FAIL; r->src_len = 0; r->dest_len = 0;
            if (dropped[i].cb) dropped[i].cb(r);
            if (dropped[i].done) { pthread_mutex_lock(&g_aq_lock); *dropped[i].done = 1; pthread_cond_broadcast(&g_aq_done); pthread_mutex_unlock(&g_aq_lock); }
        }
        return;
    }
    for (;;) {
        bool busy = false;
        for (size_t i = 0; i < g_aq_running.size() && !busy; i++) busy = !sess || g_aq_running[i] == sess;
        for (size_t i = g_aq_head; i < g_aq.size() && !busy; i++) busy = !sess || g_aq[i].sess == sess;
        if (!busy) break;
        pthread_cond_wait(&g_aq_idle, &g_aq_lock);
    }
    pthread_mutex_unlock(&g_aq_lock);
}

/* fork(): the child inherits the queue's state but not its consumer thread (the reference API supports forked workers,
 * max_forks).  Its first queued call would wait for ever: start over - empty queue, fresh lock (it may have been held at
 * the moment of the fork), no consumer. */
static void aq_atfork_child(void)
{
    pthread_mutex_init(&g_aq_lock, NULL);
    pthread_cond_init(&g_aq_more, NULL); pthread_cond_init(&g_aq_idle, NULL); pthread_cond_init(&g_aq_done, NULL);
    g_aq.clear(); g_aq_head = 0; g_aq_running.clear();
    g_aq_thread = false;
}

static bool consumer_started_locked()
{
    static bool atfork_set = false;
    if (!atfork_set) { pthread_atfork(NULL, NULL, aq_atfork_child); atfork_set = true; }
    if (!g_aq_thread) {
        pthread_t th;
        if (pthread_create(&th, NULL, async_consumer, NULL) != 0) return false;
        pthread_detach(th);
        g_aq_tid = th; g_aq_thread = true;
    }
    return true;
}

/* A synchronous qzCompress / qzDecompress of a small request does what the asynchronous API does and waits: alone it
 * would occupy ONE wave of the GPU for milliseconds (a chunk's parse is serial); queued, the calls that many threads make
 * at the same time - the reference's perf harness issues one qzCompress per block from every thread,
 * test/main.c:2175-2299, and its engine keeps <= 32 chunks in flight per instance, src/qatzip_internal.h:65-70 - share
 * launches.  Only requests the batch paths can carry whole (closed members of a deflate format; single sized gzip-ext
 * members to decode); everything else, and everything when QATZIP_AMD_SYNC_COALESCE=0, runs as its own call. */
#define AQ_SYNC_MAX_REQ (1u << 20)
static bool sync_via_queue(QzSession_T *sess, Sess *s, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                           unsigned int *dest_len, unsigned long *crc, bool compress, int *rc_out)
{
    static const bool enabled = !(getenv("QATZIP_AMD_SYNC_COALESCE") && getenv("QATZIP_AMD_SYNC_COALESCE")[0] == '0');
    if (!enabled || *src_len > AQ_SYNC_MAX_REQ || *src_len == 0) return false;
    if (g_aq_thread && pthread_equal(pthread_self(), g_aq_tid)) return false;      /* a callback calling back in */
    const int f = s->p.fmt;
    if (compress) {
        if (!(f == F_GZIP || f == F_GZIP_EXT || f == F_RAW || f == F_4B) || s->open || s->hw_framing) return false;
        if (s->p.comp_lvl < 1 || s->p.comp_lvl > 9) return false;
    } else {
        if (f != F_GZIP_EXT || s->d_hold || s->p.stop_at_stream_end) return false;
        uint32_t es = 0, ed = 0;
        const int hl = parse_header(F_GZIP_EXT, src, *src_len, &es, &ed);
        if (hl < 0 || es == 0 || ed == 0 || (uint64_t)hl + ed + 8 != *src_len || es > *dest_len || es > AQ_BATCH_MAX_REQ) return false;
    }
    QzResult_T r;
    memset(&r, 0, sizeof(r));
    r.src_len = *src_len; r.dest_len = *dest_len;
    volatile int done = 0;
    pthread_mutex_lock(&g_aq_lock);
    if (!consumer_started_locked()) { pthread_mutex_unlock(&g_aq_lock); return false; }
    AsyncReq q = { sess, src, dest, NULL, &r, compress, crc, &done };
    g_aq.push_back(q);
    pthread_cond_signal(&g_aq_more);
    while (!done) pthread_cond_wait(&g_aq_done, &g_aq_lock);
    pthread_mutex_unlock(&g_aq_lock);
    *src_len = r.src_len; *dest_len = r.dest_len;
    *rc_out = r.status;
    sess->thd_sess_stat = r.status;
    return true;
}

static int submit2(QzSession_T *sess, const unsigned char *src, unsigned char *dest, qzAsyncCallbackFn cb, QzResult_T *r, bool compress)
{
    if (!r) return QZ_PARAMS;
    if (!cb) return run_sync2(sess, src, dest, r, compress);
    if (!sess || !src || !dest) return QZ_PARAMS;
    pthread_mutex_lock(&g_aq_lock);
    if (!consumer_started_locked()) { pthread_mutex_unlock(&g_aq_lock); return QZ_FAIL; }
    AsyncReq q = { sess, src, dest, cb, r, compress, NULL, NULL };
    g_aq.push_back(q);
    pthread_cond_signal(&g_aq_more);
    pthread_mutex_unlock(&g_aq_lock);
    return QZ_OK;
}

extern "C" int qzCompress2(QzSession_T *sess, const unsigned char *src, unsigned char *dest, qzAsyncCallbackFn callback, QzResult_T *qzResults)
{ return submit2(sess, src, dest, callback, qzResults, true); }
extern "C" int qzDecompress2(QzSession_T *sess, const unsigned char *src, unsigned char *dest, qzAsyncCallbackFn callback, QzResult_T *qzResults)
{ return submit2(sess, src, dest, callback, qzResults, false); }

/* ------------------------------------------------------------------ declared-only surface */
#define NS(...) { return QZ_NOT_SUPPORTED; }
extern "C" int qzCompressCrc64(QzSession_T *, const unsigned char *, unsigned int *, unsigned char *, unsigned int *, unsigned int, uint64_t *) NS()
extern "C" int qzCompressCrc64Ext(QzSession_T *, const unsigned char *, unsigned int *, unsigned char *, unsigned int *, unsigned int, uint64_t *, uint64_t *) NS()
extern "C" int qzDecompressCrc64(QzSession_T *, const unsigned char *, unsigned int *, unsigned char *, unsigned int *, uint64_t *) NS()
extern "C" int qzDecompressCrc64Ext(QzSession_T *, const unsigned char *, unsigned int *, unsigned char *, unsigned int *, uint64_t *, uint64_t *) NS()
extern "C" int qzCompressWithMetadataExt(QzSession_T *, const unsigned char *, unsigned int *, unsigned char *, unsigned int *, unsigned int, uint64_t *, QzMetadataBlob_T, uint32_t, uint32_t) NS()
extern "C" int qzDecompressWithMetadataExt(QzSession_T *, const unsigned char *, unsigned int *, unsigned char *, unsigned int *, uint64_t *, QzMetadataBlob_T, uint32_t) NS()
extern "C" int qzAllocateMetadata(QzMetadataBlob_T *, size_t, uint32_t) NS()
extern "C" int qzFreeMetadata(QzMetadataBlob_T) NS()
extern "C" int qzMetadataBlockRead(uint32_t, QzMetadataBlob_T, uint32_t *, uint32_t *, uint32_t *, uint32_t *) NS()
extern "C" int qzMetadataBlockWrite(uint32_t, QzMetadataBlob_T, uint32_t *, uint32_t *, uint32_t *, uint32_t *) NS()
extern "C" int qzMetadataBlockGetCrc64(uint32_t, QzMetadataBlob_T, uint64_t *, uint64_t *) NS()
extern "C" int qzMetadataBlockGetCrc32(uint32_t, QzMetadataBlob_T, uint32_t *, uint32_t *) NS()
extern "C" int qzGetSessionCrc64Config(QzSession_T *, QzCrc64Config_T *) NS()
extern "C" int qzGetSessionCrc32Config(QzSession_T *, QzCrc32Config_T *) NS()
extern "C" int qzSetSessionCrc64Config(QzSession_T *, QzCrc64Config_T *) NS()
extern "C" int qzSetSessionCrc32Config(QzSession_T *, QzCrc32Config_T *) NS()
extern "C" int qzGetSoftwareComponentCount(unsigned int *n) { if (!n) return QZ_PARAMS; *n = 1; return QZ_OK; }
extern "C" int qzGetSoftwareComponentVersionList(QzSoftwareVersionInfo_T *info, unsigned int *n)
{
    if (!info || !n || *n < 1) return QZ_PARAMS;
    memset(info, 0, sizeof(*info));
    info->component_type = QZ_COMPONENT_QATZIP_API;
    snprintf((char *)info->component_name, QZ_MAX_STRING_LENGTH, "qatzip-amd (MI355X)");
    info->major_version = QATZIP_API_VERSION_NUM_MAJOR; info->minor_version = QATZIP_API_VERSION_NUM_MINOR;
    *n = 1;
    return QZ_OK;
}
