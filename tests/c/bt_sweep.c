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
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
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

int main(int argc, char **argv)
{
    if (argc >= 5 && !strcmp(argv[1], "sweep")) return sweep(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    if (argc >= 5 && !strcmp(argv[1], "perf")) return perf(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    fprintf(stderr, "usage: bt_sweep sweep <start> <end> <step> | perf <mbytes> <block> <loops>\n");
    return 64;
}
