/*
 * bt_sweep.c — drop-in proof: a plain C caller written against qatzip.h only, in the
 * spirit of the reference's boundary test (test/bt.c:133-170: for a sweep of lengths,
 * compress / decompress / compare over three corpora) and of its perf harness
 * (test/main.c:2204-2299: qzCompress(last=1) per block, qzDecompress per recorded
 * block, Gbps = bytes*8*count/2^30/s).  Links against libqatzip_amd.so exactly as
 * it would link against libqatzip.so.
 *
 *   bt_sweep sweep <start> <end> <step>       round-trip sweep, exit 0 when all equal
 *   bt_sweep perf <mbytes> <block> <loops>    per-block calls, prints comp/decomp Gbps
 *   bt_sweep perfmt <mbytes> <block> <loops> <threads>   the harness' -t: every thread its own session and its own
 *                                             <mbytes> of data, one qzCompress / qzDecompress per block, started together
 *                                             (test/main.c:2175-2202), rates summed as run_perf_test.sh:111-123 does
 *   bt_sweep run [-i file] [-t threads] [-l loops] [-v] [-C hw_buff_sz] [-b block_size] [-D comp|decomp|both]
 *                [-O deflate|gzip|gzipext|deflate_4B|lz4|zlib] [-L level] [-p pinned|common] [-s bytes]
 *                                             the harness' own option letters (test/main.c:6319-6374, its test mode 4): every
 *                                             thread sets a session up with these parameters and runs `loops` passes of
 *                                             compress (one qzCompress per block, or one call for the whole buffer when -b is
 *                                             not given) and / or decompress over the file (or -s bytes of generated runs,
 *                                             512 KB by default as in the harness); -v compares what came back
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include "qatzip.h"

static void fill(unsigned char *p, size_t n, int corpus)
{
    size_t j;
    for (j = 0; j < n; j++)
        p[j] = corpus == 0 ? (unsigned char)(j % 200) : corpus == 1 ? (unsigned char)(rand() % 255) : 'A';
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

static int sweep(unsigned start, unsigned end, unsigned step)
{
    QzSession_T sess;
    unsigned n, cap = qzMaxCompressedLength(end, NULL) + 64;
    unsigned char *src = qzMalloc(end + 1, 0, COMMON_MEM), *comp = qzMalloc(cap, 0, COMMON_MEM),
                  *back = qzMalloc(end + 1, 0, COMMON_MEM);
    int corpus, bad = 0, rc;
    memset(&sess, 0, sizeof(sess));
    if (!src || !comp || !back) return 2;
    for (corpus = 0; corpus < 3; corpus++) {
        fill(src, end, corpus);
        for (n = start; n <= end; n += step) {
            unsigned sl = n, dl = cap, cl, ol = end + 1;
            rc = qzCompress(&sess, src, &sl, comp, &dl, 1);
            if (rc != QZ_OK || sl != n) { printf("compress rc %d n %u\n", rc, n); bad++; continue; }
            cl = dl;
            rc = qzDecompress(&sess, comp, &cl, back, &ol);
            if (rc != QZ_OK || ol != n || cl != dl || memcmp(src, back, n)) { printf("roundtrip mismatch corpus %d n %u rc %d\n", corpus, n, rc); bad++; }
        }
    }
    qzTeardownSession(&sess);
    qzClose(&sess);
    qzFree(src); qzFree(comp); qzFree(back);
    printf("sweep done, %d failures\n", bad);
    return bad ? 1 : 0;
}

static int perf(unsigned mb, unsigned block, unsigned loops)
{
    QzSession_T sess;
    size_t total = (size_t)mb << 20, off, nblk = (total + block - 1) / block, i;
    const char *pe = getenv("BT_PINNED");       /* unset: source pinned, the rest pageable; 1: everything pinned; 0: nothing */
    unsigned char *src = (pe && !atoi(pe)) ? NULL : qzMalloc(total, 0, PINNED_MEM), *comp, *back;
    unsigned *csz = malloc(nblk * sizeof(unsigned)), cap = qzMaxCompressedLength(block, NULL) + 64, l;
    double t0, tc = 0, td = 0;
    memset(&sess, 0, sizeof(sess));
    if (!src) src = qzMalloc(total, 0, COMMON_MEM);
    {
        const int pin = pe && atoi(pe);
        comp = qzMalloc((size_t)cap * nblk, 0, pin ? PINNED_MEM : COMMON_MEM); back = qzMalloc(total, 0, pin ? PINNED_MEM : COMMON_MEM);
    }
    if (!src || !comp || !back || !csz) return 2;
    srand(1);
    for (off = 0; off < total;) {               /* genRandomData-style runs, test/main.c:293-310 */
        size_t run = (size_t)(rand() % 100), k; unsigned char v = (unsigned char)(rand() % 65 + 90);
        for (k = 0; k < run && off < total; k++) src[off++] = v;
    }
    qzSetLogLevel(LOG_NONE);
    for (l = 0; l < loops; l++) {
        t0 = now();
        for (i = 0, off = 0; i < nblk; i++, off += block) {
            unsigned sl = (unsigned)(total - off < block ? total - off : block), dl = cap;
            if (qzCompress(&sess, src + off, &sl, comp + (size_t)i * cap, &dl, 1) != QZ_OK) return 3;
            csz[i] = dl;
        }
        tc += now() - t0;
        t0 = now();
        for (i = 0, off = 0; i < nblk; i++, off += block) {
            unsigned cl = csz[i], ol = (unsigned)(total - off < block ? total - off : block);
            if (qzDecompress(&sess, comp + (size_t)i * cap, &cl, back + off, &ol) != QZ_OK) return 4;
        }
        td += now() - t0;
    }
    if (memcmp(src, back, total)) { printf("perf: data mismatch\n"); return 5; }
    printf("perf: %u MiB block %u loops %u  compress %.3f Gbps  decompress %.3f Gbps (host to host)\n", mb, block, loops,
           (double)total * 8 * loops / 1073741824.0 / tc, (double)total * 8 * loops / 1073741824.0 / td);
    qzTeardownSession(&sess);
    return 0;
}

/* ---- threads: one session each, synchronous calls, everybody started by a barrier ---- */
typedef struct { unsigned mb, block, loops, id; double tc, td; int rc; pthread_barrier_t *bar; } mt_arg;

static void *mt_body(void *p)
{
    mt_arg *a = (mt_arg *)p;
    QzSession_T sess;
    size_t total = (size_t)a->mb << 20, off, nblk = (total + a->block - 1) / a->block, i;
    unsigned cap = qzMaxCompressedLength(a->block, NULL) + 64, l, seed = 17 + a->id;
    unsigned char *src = qzMalloc(total, 0, COMMON_MEM), *comp = qzMalloc((size_t)cap * nblk, 0, COMMON_MEM), *back = qzMalloc(total, 0, COMMON_MEM);
    unsigned *csz = malloc(nblk * sizeof(unsigned));
    double t0;
    memset(&sess, 0, sizeof(sess));
    a->rc = 2; a->tc = a->td = 0;
    if (src && comp && back && csz) {
        for (off = 0; off < total;) {           /* genRandomData-style runs, a different stream per thread */
            size_t run = (size_t)(rand_r(&seed) % 100), k; unsigned char v = (unsigned char)(rand_r(&seed) % 65 + 90);
            for (k = 0; k < run && off < total; k++) src[off++] = v;
        }
        a->rc = 0;
        {   /* the session (and its device context) exists before the clock starts */
            unsigned sl = a->block < total ? a->block : (unsigned)total, dl = cap;
            if (qzCompress(&sess, src, &sl, comp, &dl, 1) != QZ_OK) a->rc = 3;
        }
    }
    pthread_barrier_wait(a->bar);
    for (l = 0; l < a->loops; l++) {             /* a thread that failed still keeps the barriers' count */
        t0 = now();
        for (i = 0, off = 0; i < nblk && a->rc == 0; i++, off += a->block) {
            unsigned sl = (unsigned)(total - off < a->block ? total - off : a->block), dl = cap;
            if (qzCompress(&sess, src + off, &sl, comp + (size_t)i * cap, &dl, 1) != QZ_OK) a->rc = 3;
            csz[i] = dl;
        }
        a->tc += now() - t0;
        pthread_barrier_wait(a->bar);
        t0 = now();
        for (i = 0, off = 0; i < nblk && a->rc == 0; i++, off += a->block) {
            unsigned cl = csz[i], ol = (unsigned)(total - off < a->block ? total - off : a->block);
            if (qzDecompress(&sess, comp + (size_t)i * cap, &cl, back + off, &ol) != QZ_OK) a->rc = 4;
        }
        a->td += now() - t0;
        pthread_barrier_wait(a->bar);
    }
    if (a->rc == 0 && memcmp(src, back, total)) a->rc = 5;
    qzTeardownSession(&sess);
    qzFree(src); qzFree(comp); qzFree(back); free(csz);
    return NULL;
}

static int perfmt(unsigned mb, unsigned block, unsigned loops, unsigned nthr)
{
    pthread_t th[256]; mt_arg arg[256]; pthread_barrier_t bar;
    unsigned t; double gc = 0, gd = 0; int bad = 0;
    if (nthr < 1 || nthr > 256) return 64;
    qzSetLogLevel(LOG_NONE);
    pthread_barrier_init(&bar, NULL, nthr);
    for (t = 0; t < nthr; t++) { arg[t].mb = mb; arg[t].block = block; arg[t].loops = loops; arg[t].id = t; arg[t].bar = &bar; pthread_create(&th[t], NULL, mt_body, &arg[t]); }
    for (t = 0; t < nthr; t++) {
        pthread_join(th[t], NULL);
        if (arg[t].rc) { printf("perfmt: thread %u failed (%d)\n", t, arg[t].rc); bad = 1; continue; }
        gc += (double)mb * 1048576.0 * 8 * loops / 1073741824.0 / arg[t].tc;
        gd += (double)mb * 1048576.0 * 8 * loops / 1073741824.0 / arg[t].td;
    }
    printf("perfmt: %u thread(s) x %u MiB, block %u, loops %u  compress %.3f Gbps  decompress %.3f Gbps (sum over threads, host to host)\n",
           nthr, mb, block, loops, gc, gd);
    return bad ? 1 : 0;
}

/* ---- run: the reference harness' option letters ---- */
#include <unistd.h>
typedef struct {
    const unsigned char *data; size_t size; unsigned hw, block, loops, level, id; int dir, fmt, verify, pinned;
    double secs; int rc; pthread_barrier_t *bar;
} run_arg;
enum { RUN_BOTH, RUN_COMP, RUN_DECOMP };
enum { FMT_RAW, FMT_GZIP, FMT_GZIPEXT, FMT_4B, FMT_LZ4, FMT_ZLIB };

static int run_setup(QzSession_T *sess, const run_arg *a)
{
    if (a->fmt == FMT_LZ4) {
        QzSessionParamsLZ4_T p;
        if (qzGetDefaultsLZ4(&p) != QZ_OK) return -1;
        p.common_params.hw_buff_sz = a->hw; p.common_params.comp_lvl = a->level;
        return qzSetupSessionLZ4(sess, &p);
    }
    if (a->fmt == FMT_ZLIB) {
        QzSessionParamsDeflateExt_T p;
        if (qzGetDefaultsDeflateExt(&p) != QZ_OK) return -1;
        p.deflate_params.data_fmt = QZ_DEFLATE_RAW; p.zlib_format = 1;
        p.deflate_params.common_params.hw_buff_sz = a->hw; p.deflate_params.common_params.comp_lvl = a->level;
        return qzSetupSessionDeflateExt(sess, &p);
    }
    {
        QzSessionParamsDeflate_T p;
        if (qzGetDefaultsDeflate(&p) != QZ_OK) return -1;
        p.data_fmt = a->fmt == FMT_RAW ? QZ_DEFLATE_RAW : a->fmt == FMT_GZIP ? QZ_DEFLATE_GZIP : a->fmt == FMT_4B ? QZ_DEFLATE_4B : QZ_DEFLATE_GZIP_EXT;
        p.common_params.hw_buff_sz = a->hw; p.common_params.comp_lvl = a->level;
        return qzSetupSessionDeflate(sess, &p);
    }
}

static void *run_body(void *vp)
{
    run_arg *a = (run_arg *)vp;
    QzSession_T sess;
    const size_t block = a->block ? a->block : a->size, nblk = a->size ? (a->size + block - 1) / block : 1;
    size_t i, off, cap_total = 0;
    unsigned *csz = malloc(nblk * sizeof(unsigned)), *cof = malloc(nblk * sizeof(unsigned)), l;
    unsigned char *src = NULL, *comp = NULL, *back = NULL;
    double t0;
    memset(&sess, 0, sizeof(sess));
    a->rc = 2; a->secs = 0;
    if (run_setup(&sess, a) >= 0 && csz && cof) {
        const unsigned cap = qzMaxCompressedLength(block < a->size ? block : a->size, &sess) + 64;
        cap_total = (size_t)cap * nblk;
        src = qzMalloc(a->size ? a->size : 1, 0, a->pinned ? PINNED_MEM : COMMON_MEM);
        comp = qzMalloc(cap_total, 0, a->pinned ? PINNED_MEM : COMMON_MEM);
        back = qzMalloc(a->size ? a->size : 1, 0, a->pinned ? PINNED_MEM : COMMON_MEM);
        if (src && comp && back) {
            memcpy(src, a->data, a->size);
            a->rc = 0;
            /* -D decomp needs something to decompress: one untimed compress pass (the harness does the same) */
            for (i = 0, off = 0; i < nblk && a->rc == 0; i++, off += block) {
                unsigned sl = (unsigned)(a->size - off < block ? a->size - off : block), dl = cap;
                cof[i] = (unsigned)(i * cap);
                if (qzCompress(&sess, src + off, &sl, comp + cof[i], &dl, 1) != QZ_OK) a->rc = 3;
                csz[i] = dl;
            }
        }
    }
    pthread_barrier_wait(a->bar);
    t0 = now();
    for (l = 0; l < a->loops && a->rc == 0; l++) {
        if (a->dir != RUN_DECOMP)
            for (i = 0, off = 0; i < nblk && a->rc == 0; i++, off += block) {
                unsigned sl = (unsigned)(a->size - off < block ? a->size - off : block), dl = (unsigned)(cap_total / nblk);
                if (qzCompress(&sess, src + off, &sl, comp + cof[i], &dl, 1) != QZ_OK) a->rc = 3;
                csz[i] = dl;
            }
        if (a->dir != RUN_COMP)
            for (i = 0, off = 0; i < nblk && a->rc == 0; i++, off += block) {
                unsigned cl = csz[i], ol = (unsigned)(a->size - off < block ? a->size - off : block);
                if (qzDecompress(&sess, comp + cof[i], &cl, back + off, &ol) != QZ_OK) a->rc = 4;
            }
        if (a->verify && a->dir != RUN_COMP && a->rc == 0 && memcmp(src, back, a->size)) a->rc = 5;
    }
    a->secs = now() - t0;
    qzTeardownSession(&sess);
    qzFree(src); qzFree(comp); qzFree(back); free(csz); free(cof);
    return NULL;
}

static int run(int argc, char **argv)
{
    run_arg proto; pthread_t th[256]; run_arg arg[256]; pthread_barrier_t bar;
    const char *file = NULL; unsigned nthr = 1, t; size_t gen = 512 * 1024; unsigned char *data; int c, bad = 0; double sum = 0;
    memset(&proto, 0, sizeof(proto));
    proto.hw = 64 * 1024; proto.block = 0; proto.loops = 2; proto.level = 1; proto.dir = RUN_BOTH; proto.fmt = FMT_GZIPEXT;
    optind = 2;
    while ((c = getopt(argc, argv, "i:t:l:vC:b:D:O:L:p:s:")) != -1) {
        switch (c) {
        case 'i': file = optarg; break;
        case 't': nthr = (unsigned)atoi(optarg); break;
        case 'l': proto.loops = (unsigned)atoi(optarg); break;
        case 'v': proto.verify = 1; break;
        case 'C': proto.hw = (unsigned)atoi(optarg); break;
        case 'b': proto.block = (unsigned)atoi(optarg); break;
        case 'L': proto.level = (unsigned)atoi(optarg); break;
        case 's': gen = (size_t)atol(optarg); break;
        case 'p': proto.pinned = !strcmp(optarg, "pinned"); break;
        case 'D': proto.dir = !strcmp(optarg, "comp") ? RUN_COMP : !strcmp(optarg, "decomp") ? RUN_DECOMP : RUN_BOTH; break;
        case 'O': proto.fmt = !strcmp(optarg, "deflate") ? FMT_RAW : !strcmp(optarg, "gzip") ? FMT_GZIP : !strcmp(optarg, "deflate_4B") ? FMT_4B :
                              !strcmp(optarg, "lz4") ? FMT_LZ4 : !strcmp(optarg, "zlib") ? FMT_ZLIB : FMT_GZIPEXT; break;
        default: return 64;
        }
    }
    if (nthr < 1 || nthr > 256 || proto.loops < 1) return 64;
    if (file) {
        FILE *f = fopen(file, "rb"); long sz;
        if (!f) { perror(file); return 66; }
        fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET);
        data = malloc(sz > 0 ? (size_t)sz : 1);
        if (!data || fread(data, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return 66; }
        fclose(f); gen = (size_t)sz;
    } else {
        size_t off; unsigned seed = 7;
        data = malloc(gen ? gen : 1);
        if (!data) return 2;
        for (off = 0; off < gen;) {              /* genRandomData-style runs, test/main.c:293-310 */
            size_t runl = (size_t)(rand_r(&seed) % 100), k; unsigned char v = (unsigned char)(rand_r(&seed) % 65 + 90);
            for (k = 0; k < runl && off < gen; k++) data[off++] = v;
        }
    }
    qzSetLogLevel(LOG_NONE);
    pthread_barrier_init(&bar, NULL, nthr);
    for (t = 0; t < nthr; t++) { arg[t] = proto; arg[t].data = data; arg[t].size = gen; arg[t].id = t; arg[t].bar = &bar; pthread_create(&th[t], NULL, run_body, &arg[t]); }
    for (t = 0; t < nthr; t++) {
        pthread_join(th[t], NULL);
        if (arg[t].rc) { printf("run: thread %u failed (%d)\n", t, arg[t].rc); bad = 1; continue; }
        /* the harness counts the source bytes once per direction (test/main.c:2336-2346) */
        sum += (double)gen * 8 * arg[t].loops * (proto.dir == RUN_BOTH ? 2 : 1) / 1073741824.0 / arg[t].secs;
    }
    printf("run: %u thread(s), %zu bytes, hw_buff_sz %u, block %u, loops %u, %s%s: %.3f Gbps (sum over threads, host to host)\n",
           nthr, gen, proto.hw, proto.block, proto.loops, proto.dir == RUN_COMP ? "comp" : proto.dir == RUN_DECOMP ? "decomp" : "both",
           proto.verify ? ", verified" : "", sum);
    free(data);
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && !strcmp(argv[1], "run")) return run(argc, argv);
    if (argc >= 6 && !strcmp(argv[1], "perfmt")) return perfmt(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
    if (argc >= 5 && !strcmp(argv[1], "sweep")) return sweep(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    if (argc >= 5 && !strcmp(argv[1], "perf")) return perf(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    fprintf(stderr, "usage: bt_sweep sweep <start> <end> <step> | perf <mbytes> <block> <loops> | perfmt <mbytes> <block> <loops> <threads> |\n"
                    "       run [-i file] [-t threads] [-l loops] [-v] [-C hw_buff_sz] [-b block_size] [-D comp|decomp|both]\n"
                    "           [-O deflate|gzip|gzipext|deflate_4B|lz4|zlib] [-L level] [-p pinned|common] [-s bytes]\n");
    return 64;
}
