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

int main(int argc, char **argv)
{
    if (argc >= 6 && !strcmp(argv[1], "perfmt")) return perfmt(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
    if (argc >= 5 && !strcmp(argv[1], "sweep")) return sweep(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    if (argc >= 5 && !strcmp(argv[1], "perf")) return perf(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    fprintf(stderr, "usage: bt_sweep sweep <start> <end> <step> | perf <mbytes> <block> <loops>\n");
    return 64;
}
