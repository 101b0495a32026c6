/*
 * qzip-amd — file front end of the MI355X backend, the counterpart of the reference's
 * `qzip` utility (utils/qzip_main.c:53-330 option handling, utils/qzip.c:180-404
 * slab loop, :659-771 naming rules).  Written against qatzip.h only: it links with
 * libqatzip_amd.so exactly as the reference tool links with libqatzip.so.
 *
 * Same observable behaviour for the formats this backend implements:
 *   compress   file -> file.gz (gzip / gzipext / deflate_4B) or file.lz4; the input is
 *              read in slabs of <= 512 MiB and every `-b` block becomes one complete
 *              member / frame (qzCompress(..., last = 1)), so `gzip -d` reads the result;
 *   decompress file.gz / file.lz4 -> file; members may straddle slab boundaries;
 *              the destination grows by the reference's ratios (5, 20, 50, 100);
 *   no file arguments: stdin -> stdout;  -R: walk directories;  -k: keep the input;
 *   the output inherits the input's modification time.
 * Not offered (QZ_NOT_SUPPORTED in the library): 7z archives (-O 7z), lz4s, zstd.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
#include "qatzip.h"

#define SLAB_BYTES (512u * 1024 * 1024)

enum { FMT_GZIP, FMT_GZIPEXT, FMT_4B, FMT_LZ4 };

static struct {
    int decompress, keep, force, recursive, fmt, fmt_given, quiet;
    unsigned level, chunk, block;
    const char *out_name, *prog;
} g = { 0, 0, 0, 0, FMT_GZIPEXT, 0, 0, 1, 65536, SLAB_BYTES, NULL, "qzip-amd" };

static const unsigned grow_ratio[] = { 5, 20, 50, 100 };

static void usage(FILE *f)
{
    fprintf(f,
            "Usage: %s [options] [file...]        (no file: stdin -> stdout)\n"
            "  -d            decompress\n"
            "  -k            keep (do not delete) the input file\n"
            "  -f            force: write compressed data to a terminal\n"
            "  -R            operate recursively on directories\n"
            "  -o <name>     output name (the format suffix is appended when compressing)\n"
            "  -A <alg>      deflate | lz4                         (default deflate)\n"
            "  -O <fmt>      gzip | gzipext | deflate_4B | lz4     (default gzipext)\n"
            "  -L <level>    compression level                     (deflate: 1-9, zlib's levels; lz4: 1-2)\n"
            "  -C <bytes>    chunk size, a power of two 1K..512K   (default 65536)\n"
            "  -b <bytes>    bytes per qzCompress call = per member (default 512 MiB)\n"
            "  -q            no statistics\n"
            "  -h            this text\n", g.prog);
}

static double now_s(void)
{
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + t.tv_usec * 1e-6;
}

static const char *suffix_of_fmt(void) { return g.fmt == FMT_LZ4 ? ".lz4" : ".gz"; }

static int ends_with(const char *s, const char *suf)
{
    size_t a = strlen(s), b = strlen(suf);
    return a >= b && strcmp(s + a - b, suf) == 0;
}

static int setup_session(QzSession_T *sess)
{
    int rc;
    memset(sess, 0, sizeof(*sess));
    if (g.fmt == FMT_LZ4) {
        QzSessionParamsLZ4_T p;
        if ((rc = qzGetDefaultsLZ4(&p)) < 0) return rc;
        p.common_params.comp_lvl = g.level;
        p.common_params.hw_buff_sz = g.chunk;
        p.common_params.direction = QZ_DIR_BOTH;
        return qzSetupSessionLZ4(sess, &p);
    } else {
        QzSessionParamsDeflate_T p;
        if ((rc = qzGetDefaultsDeflate(&p)) < 0) return rc;
        p.data_fmt = g.fmt == FMT_GZIP ? QZ_DEFLATE_GZIP : g.fmt == FMT_4B ? QZ_DEFLATE_4B : QZ_DEFLATE_GZIP_EXT;
        p.common_params.comp_lvl = g.level;
        p.common_params.hw_buff_sz = g.chunk;
        p.common_params.direction = QZ_DIR_BOTH;
        return qzSetupSessionDeflate(sess, &p);
    }
}

static size_t read_full(FILE *f, unsigned char *p, size_t n)
{
    size_t got = 0;
    while (got < n) {
        size_t r = fread(p + got, 1, n - got, f);
        if (r == 0) break;
        got += r;
    }
    return got;
}

/* compress everything `in` yields: one member per block of <= g.block bytes */
static int compress_stream(QzSession_T *sess, FILE *in, FILE *out, unsigned long long *n_in, unsigned long long *n_out)
{
    const unsigned slab = g.block < SLAB_BYTES ? (g.block > (1u << 20) ? SLAB_BYTES : 64u << 20) : SLAB_BYTES;
    unsigned char *src = qzMalloc(slab, 0, PINNED_MEM), *dst;
    unsigned cap = qzMaxCompressedLength(g.block < slab ? g.block : slab, sess) + 64;
    int first = 1, rc = QZ_OK;
    if (!src) src = qzMalloc(slab, 0, COMMON_MEM);
    dst = qzMalloc(cap, 0, COMMON_MEM);
    if (!src || !dst) { fprintf(stderr, "%s: out of memory\n", g.prog); return QZ_FAIL; }
    for (;;) {
        size_t got = read_full(in, src, slab), off = 0;
        if (got == 0 && !first) break;
        do {                                            /* an empty input still yields one (empty) member */
            unsigned sl = (unsigned)(got - off < g.block ? got - off : g.block), dl = cap;
            rc = qzCompress(sess, src + off, &sl, dst, &dl, 1);
            if (rc != QZ_OK) { fprintf(stderr, "%s: compression failed: %d\n", g.prog, rc); goto done; }
            if (fwrite(dst, 1, dl, out) != dl) { perror("write"); rc = QZ_FAIL; goto done; }
            off += sl; *n_in += sl; *n_out += dl;
        } while (off < got);
        first = 0;
        if (got < slab) break;
    }
done:
    qzFree(src); qzFree(dst);
    return rc;
}

/* decompress a concatenation of members / frames; a member may continue in the next slab */
static int decompress_stream(QzSession_T *sess, FILE *in, FILE *out, unsigned long long *n_in, unsigned long long *n_out)
{
    const unsigned slab = 256u << 20;
    unsigned char *src = qzMalloc(slab, 0, PINNED_MEM), *dst = NULL;
    unsigned ratio_idx = 0, have = 0;
    unsigned long long dcap = 0;
    int rc = QZ_OK, eof = 0;
    if (!src) src = qzMalloc(slab, 0, COMMON_MEM);
    if (!src) { fprintf(stderr, "%s: out of memory\n", g.prog); return QZ_FAIL; }
    for (;;) {
        if (!eof && have < slab) {
            size_t got = read_full(in, src + have, slab - have);
            if (got < slab - have) eof = 1;
            have += (unsigned)got;
        }
        if (have == 0) break;
        if (!dst) {
            dcap = (unsigned long long)have * grow_ratio[ratio_idx];
            if (dcap < (1u << 20)) dcap = 1u << 20;
            if (dcap > 0xfff00000ull) dcap = 0xfff00000ull;
            dst = qzMalloc((size_t)dcap, 0, COMMON_MEM);
            if (!dst) { fprintf(stderr, "%s: out of memory\n", g.prog); rc = QZ_FAIL; break; }
        }
        unsigned sl = have, dl = (unsigned)dcap;
        rc = qzDecompress(sess, src, &sl, dst, &dl);
        if (rc == QZ_OK || (rc == QZ_BUF_ERROR && sl > 0)) {
            if (dl && fwrite(dst, 1, dl, out) != dl) { perror("write"); rc = QZ_FAIL; break; }
            *n_in += sl; *n_out += dl;
            memmove(src, src + sl, have - sl);
            have -= sl;
            if (sl == 0 && dl == 0 && eof) { fprintf(stderr, "%s: unexpected end of input\n", g.prog); rc = QZ_DATA_ERROR; break; }
            if (sl == 0 && dl == 0 && have == slab) { fprintf(stderr, "%s: a member larger than %u bytes of input is not supported\n", g.prog, slab); rc = QZ_FAIL; break; }
            rc = QZ_OK;
            continue;
        }
        if (rc == QZ_BUF_ERROR) {                       /* nothing fitted: grow the destination like the reference */
            if (++ratio_idx >= sizeof(grow_ratio) / sizeof(grow_ratio[0]) || dcap >= 0xfff00000ull) {
                fprintf(stderr, "%s: could not expand the destination buffer any further\n", g.prog);
                break;
            }
            qzFree(dst); dst = NULL;
            continue;
        }
        if (rc == QZ_DATA_ERROR && !eof && have < slab) continue;   /* the only member so far is cut by the slab: read on */
        fprintf(stderr, "%s: decompression failed: %d\n", g.prog, rc);
        break;
    }
    qzFree(src); qzFree(dst);
    return rc;
}

static void report(const char *what, double secs, unsigned long long n_in, unsigned long long n_out)
{
    if (g.quiet) return;
    printf("%s: %llu -> %llu bytes in %.3f s", what, n_in, n_out, secs);
    if (secs > 0) printf("  (%.2f MB/s", (g.decompress ? n_out : n_in) / secs / 1e6);
    if (secs > 0 && n_in) printf(", ratio %.3f)", (double)(g.decompress ? n_in : n_out) / (double)(g.decompress ? n_out : n_in));
    else if (secs > 0) printf(")");
    printf("\n");
}

static int process_file(QzSession_T *sess, const char *in_name);

static int process_dir(QzSession_T *sess, const char *dir_name)
{
    DIR *d = opendir(dir_name);
    struct dirent *e;
    int rc = 0;
    if (!d) { perror(dir_name); return 1; }
    while ((e = readdir(d))) {
        char path[4096];
        if (e->d_name[0] == '.') continue;              /* ".", ".." and hidden files, like the reference */
        if (snprintf(path, sizeof(path), "%s/%s", dir_name, e->d_name) >= (int)sizeof(path)) { rc = 1; continue; }
        rc |= process_file(sess, path);
    }
    closedir(d);
    return rc;
}

static int process_file(QzSession_T *sess, const char *in_name)
{
    struct stat st;
    char oname[4096];
    FILE *in, *out;
    unsigned long long n_in = 0, n_out = 0;
    double t0;
    int rc;
    if (stat(in_name, &st)) { perror(in_name); return 1; }
    if (S_ISDIR(st.st_mode)) {
        if (!g.recursive) { fprintf(stderr, "%s: %s is a directory -- ignored (use -R)\n", g.prog, in_name); return 1; }
        return process_dir(sess, in_name);
    }
    if (!g.decompress) {
        if (ends_with(in_name, ".gz") || ends_with(in_name, ".lz4")) {
            fprintf(stderr, "%s: %s already has a suffix -- unchanged\n", g.prog, in_name);
            return 1;
        }
        snprintf(oname, sizeof(oname), "%s%s", g.out_name ? g.out_name : in_name, suffix_of_fmt());
    } else {
        if (!ends_with(in_name, suffix_of_fmt())) {
            fprintf(stderr, "%s: %s: wrong suffix for the selected format (%s expected)\n", g.prog, in_name, suffix_of_fmt());
            return 1;
        }
        if (g.out_name) snprintf(oname, sizeof(oname), "%s", g.out_name);
        else snprintf(oname, sizeof(oname), "%.*s", (int)(strlen(in_name) - strlen(suffix_of_fmt())), in_name);
    }
    in = fopen(in_name, "rb");
    if (!in) { perror(in_name); return 1; }
    out = fopen(oname, "wb");
    if (!out) { perror(oname); fclose(in); return 1; }
    t0 = now_s();
    rc = g.decompress ? decompress_stream(sess, in, out, &n_in, &n_out) : compress_stream(sess, in, out, &n_in, &n_out);
    fclose(in);
    if (fclose(out)) { perror(oname); rc = QZ_FAIL; }
    if (rc != QZ_OK) { unlink(oname); return 1; }
    report(g.decompress ? "decompressed" : "compressed", now_s() - t0, n_in, n_out);
    {
        struct timespec tb[2];
        memset(tb, 0, sizeof(tb));
        tb[0].tv_nsec = UTIME_NOW; tb[1].tv_sec = st.st_mtime;
        utimensat(AT_FDCWD, oname, tb, 0);
    }
    if (!g.keep) unlink(in_name);
    return 0;
}

int main(int argc, char **argv)
{
    QzSession_T sess;
    int c, rc = 0;
    char *stop;
    if (strrchr(argv[0], '/')) g.prog = strrchr(argv[0], '/') + 1; else g.prog = argv[0];
    while ((c = getopt(argc, argv, "dkfRqhVo:A:O:L:C:b:")) != -1) {
        switch (c) {
        case 'd': g.decompress = 1; break;
        case 'k': g.keep = 1; break;
        case 'f': g.force = 1; break;
        case 'R': g.recursive = 1; break;
        case 'q': g.quiet = 1; break;
        case 'h': usage(stdout); return 0;
        case 'V': printf("%s (qatzip-amd, MI355X backend)\n", g.prog); return 0;
        case 'o': g.out_name = optarg; break;
        case 'A':
            if (!strcmp(optarg, "deflate")) { if (g.fmt == FMT_LZ4) g.fmt = FMT_GZIPEXT; }
            else if (!strcmp(optarg, "lz4")) g.fmt = FMT_LZ4;
            else { fprintf(stderr, "%s: algorithm %s is not offered by this backend\n", g.prog, optarg); return 2; }
            break;
        case 'O':
            if (!strcmp(optarg, "gzip")) g.fmt = FMT_GZIP;
            else if (!strcmp(optarg, "gzipext")) g.fmt = FMT_GZIPEXT;
            else if (!strcmp(optarg, "deflate_4B")) g.fmt = FMT_4B;
            else if (!strcmp(optarg, "lz4")) g.fmt = FMT_LZ4;
            else { fprintf(stderr, "%s: format %s is not offered by this backend\n", g.prog, optarg); return 2; }
            g.fmt_given = 1;
            break;
        case 'L':
            g.level = (unsigned)strtoul(optarg, &stop, 0);
            if (*stop || g.level == 0 || g.level > 12) { fprintf(stderr, "%s: bad level %s\n", g.prog, optarg); return 2; }
            break;
        case 'C':
            g.chunk = (unsigned)strtoul(optarg, &stop, 0);
            if (*stop || g.chunk < 1024 || g.chunk > 512 * 1024 || (g.chunk & (g.chunk - 1))) { fprintf(stderr, "%s: bad chunk size %s\n", g.prog, optarg); return 2; }
            break;
        case 'b':
            g.block = (unsigned)strtoul(optarg, &stop, 0);
            if (*stop || g.block == 0 || g.block > SLAB_BYTES) { fprintf(stderr, "%s: bad block size %s\n", g.prog, optarg); return 2; }
            break;
        default: usage(stderr); return 2;
        }
    }
    qzSetLogLevel(LOG_NONE);
    /* one LZ4 frame of <= 64 KB content per call: every frame is one wave of work (a larger call would be ONE linked frame, a serial chain) */
    if (g.fmt == FMT_LZ4 && g.block > 65536) g.block = 65536;
    if (optind == argc) {                               /* stdin -> stdout */
        unsigned long long n_in = 0, n_out = 0;
        if (!g.decompress && !g.force && isatty(fileno(stdout))) {
            fprintf(stderr, "%s: compressed data not written to a terminal. Use -f to force compression.\n", g.prog);
            return 1;
        }
        if (isatty(fileno(stdin))) { usage(stdout); return 0; }
        if ((rc = setup_session(&sess)) < 0) { fprintf(stderr, "%s: session setup failed: %d\n", g.prog, rc); return 1; }
        rc = g.decompress ? decompress_stream(&sess, stdin, stdout, &n_in, &n_out) : compress_stream(&sess, stdin, stdout, &n_in, &n_out);
        fflush(stdout);
        qzTeardownSession(&sess); qzClose(&sess);
        return rc == QZ_OK ? 0 : 1;
    }
    for (; optind < argc; optind++) {
        const char *name = argv[optind];
        if (g.decompress && !g.fmt_given) {             /* the suffix picks the format, like checkSuffix() */
            if (ends_with(name, ".lz4")) g.fmt = FMT_LZ4;
            else if (ends_with(name, ".gz") && g.fmt == FMT_LZ4) g.fmt = FMT_GZIPEXT;
        }
        if (g.fmt == FMT_LZ4 && g.block > 65536) g.block = 65536;
        if ((c = setup_session(&sess)) < 0) { fprintf(stderr, "%s: session setup failed: %d\n", g.prog, c); return 1; }
        rc |= process_file(&sess, name);
        qzTeardownSession(&sess);
    }
    qzClose(&sess);
    return rc ? 1 : 0;
}
