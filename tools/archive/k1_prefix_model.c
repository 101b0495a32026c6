// CPU model of the prefix-finalising wide-window parse (no slide handling: stats only, but verified against a serial parse)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define MAXDIST 32506
#define NICE 8
#define MAXINS 4
static uint8_t d[65536 + 300];
static int n;
static inline int hash3(int p) { return ((d[p] << 12) ^ (d[p + 1] << 6) ^ d[p + 2]) & 0xffff; }
static int matchlen(int p, int q) { int ml = n - p < 258 ? n - p : 258, l = 0; while (l < ml && d[p + l] == d[q + l]) l++; return l; }

// serial reference
static int ref_len[65536], ref_dist[65536], ref_pp[65536], ref_np;
static void serial(void)
{
    static int head[65536], prev[65536];
    memset(head, 0, sizeof head);
    int p = 0; ref_np = 0;
    while (p < n) {
        int best = 2, bq = 0;
        if (n - p >= 3) {
            int h = hash3(p), q = head[h];
            prev[p] = q; head[h] = p;
            int chain = 4, lim = p > MAXDIST ? p - MAXDIST : 0;
            if (q != 0 && p - q <= MAXDIST) {
                int nice = n - p < NICE ? n - p : NICE;
                do {
                    int l = matchlen(p, q);
                    if (l > best) { best = l; bq = q; if (l >= nice) break; }
                } while ((q = prev[q]) > lim && --chain != 0);
            }
        }
        int ml = best >= 3 ? best : 0;
        ref_pp[ref_np++] = p; ref_len[p] = ml; ref_dist[p] = ml ? p - bq : 0;
        if (ml) {
            if (ml <= MAXINS && (n - p) - ml >= 3) {
                for (int k = 1; k < ml; k++) { int h = hash3(p + k); prev[p + k] = head[h]; head[h] = p + k; }
            }
            p += ml;
        } else p++;
    }
}

int W = 1024, RMAX = 3; double THETA = 0.75; int PRECISE = 1;
static int head[65536], prevt[65536];
static long st_windows, st_rounds, st_pos, st_full, st_hist[16], st_sel;

int main(int argc, char **argv)
{
    const char *file = argv[1];
    if (argc > 2) W = atoi(argv[2]);
    if (argc > 3) RMAX = atoi(argv[3]);
    if (argc > 4) THETA = atof(argv[4]);
    if (argc > 5) PRECISE = atoi(argv[5]);
    FILE *f = fopen(file, "rb");
    static uint8_t all[1 << 20];
    size_t tot = fread(all, 1, sizeof all, f);
    int nch = tot / 65536;
    int bad = 0;
    for (int c = 0; c < nch; c++) {
        n = 65536; memcpy(d, all + c * 65536, n); memset(d + n, 0, 300);
        serial();
        memset(head, 0, sizeof head);
        int ws = 0, ri = 0;  // ri: index into ref_pp for verification
        static int H[2048], prevW[2048], cpre[2048][4], ncpre[2048];
        static int LA[2048][4], nLA[2048], kex[2048], ml[2048], mq[2048], A[2048], I[2048], pp[2048], LI[2048][4], nLI[2048];
        while (ws < n) {
            int wn = n - ws < W ? n - ws : W;
            int lim = wn < W - 3 ? wn : W - 3;
            // hashes, window-local predecessor chains, pre-window candidates
            static int lasth[65536]; static int stamp[65536]; static int epoch = 0; epoch++;
            for (int t = 0; t < wn; t++) {
                int p = ws + t;
                if (n - p >= 3) {
                    int h = hash3(p); H[t] = h;
                    prevW[t] = stamp[h] == epoch ? lasth[h] : -1;
                    lasth[h] = t; stamp[h] = epoch;
                    int q = head[h], k = 0, lo = p > MAXDIST ? p - MAXDIST : 0;
                    if (q != 0 && p - q <= MAXDIST) {
                        cpre[t][k++] = q;
                        while (k < 4 && (q = prevt[q]) > lo) cpre[t][k++] = q;
                    }
                    ncpre[t] = k;
                } else { H[t] = -1; prevW[t] = -1; ncpre[t] = 0; }
                A[t] = H[t] >= 0;
            }
            int round = 0, x = -1, exitp = 0;
            int first = 1;
            for (;;) {
                // selection under A
                for (int t = 0; t < wn; t++) {
                    int k = 0;
                    if (H[t] >= 0) {
                        int q = prevW[t]; int stop = 0;
                        while (q >= 0 && k < 4) { if (A[q]) { if (ws + q == 0) { stop = 1; break; } LA[t][k++] = ws + q; } q = prevW[q]; }
                        if (!stop) {
                            // zlib: chained candidates must be > limit; first must be dist<=MAXDIST: in-window ones always are
                            for (int i = 0; i < ncpre[t] && k < 4; i++) {
                                int cq = cpre[t][i];
                                // if there were in-window candidates before, chained rule q > lo applies (already ensured for i>0; for i==0 need cq > lo)
                                int p = ws + t, lo = p > MAXDIST ? p - MAXDIST : 0;
                                if (k > 0 && cq <= lo) break;
                                LA[t][k++] = cq;
                            }
                        }
                    }
                    nLA[t] = k;
                }
                st_sel++;
                // matches (only where list changed vs previous round; model: recompute all)
                for (int t = 0; t < wn; t++) {
                    int p = ws + t, best = 2, bq = 0, ke = 0;
                    int nice = n - p < NICE ? n - p : NICE;
                    for (int i = 0; i < nLA[t]; i++) {
                        int l = matchlen(p, LA[t][i]); ke = i + 1;
                        if (l > best) { best = l; bq = LA[t][i]; if (l >= nice) break; }
                    }
                    ml[t] = best >= 3 ? best : 0; mq[t] = bq; kex[t] = ke;
                }
                // parse from 0
                memset(pp, 0, sizeof(int) * wn); memset(I, 0, sizeof(int) * wn);
                int t = 0, lastpp = 0;
                while (t < lim) {
                    pp[t] = 1; lastpp = t;
                    if (H[t] >= 0) I[t] = 1;
                    int m = ml[t];
                    if (m && m <= MAXINS && (n - (ws + t)) - m >= 3) for (int k = 1; k < m; k++) I[t + k] = 1;
                    t += m ? m : 1;
                }
                exitp = t;
                round++;
                // selection under I for parse points; find first changed
                x = -1;
                for (int t2 = 0; t2 < lim; t2++) {
                    if (!pp[t2]) continue;
                    int k = 0;
                    if (H[t2] >= 0) {
                        int q = prevW[t2]; int stop = 0;
                        while (q >= 0 && k < 4) { if (I[q]) { if (ws + q == 0) { stop = 1; break; } LI[t2][k++] = ws + q; } q = prevW[q]; }
                        if (!stop) for (int i = 0; i < ncpre[t2] && k < 4; i++) {
                            int cq = cpre[t2][i]; int p = ws + t2, lo = p > MAXDIST ? p - MAXDIST : 0;
                            if (k > 0 && cq <= lo) break;
                            LI[t2][k++] = cq;
                        }
                    }
                    nLI[t2] = k;
                    int diff = 0;
                    if (PRECISE) {
                        // same outcome if the examined prefix is identical (and, if all examined, the same count)
                        int ke = kex[t2];
                        int full = (ke == nLA[t2]) && !(ml[t2] >= (n - (ws + t2) < NICE ? n - (ws + t2) : NICE));
                        if (k < ke) diff = 1;
                        else { for (int i = 0; i < ke; i++) if (LI[t2][i] != LA[t2][i]) diff = 1; if (!diff && full && k != nLA[t2]) diff = 1; }
                    } else {
                        if (k != nLA[t2]) diff = 1; else for (int i = 0; i < k; i++) if (LI[t2][i] != LA[t2][i]) diff = 1;
                    }
                    if (diff) { x = t2; break; }
                }
                int prefix = x < 0 ? exitp : x;
                if (x < 0 || round >= RMAX || prefix >= THETA * wn) break;
                memcpy(A, I, sizeof(int) * wn);
                first = 0;
            }
            (void)first;
            int endp = x < 0 ? exitp : x;       // next window start (window-relative)
            // finalise parse points < endp: verify + commit
            for (int t = 0; t < endp && t < wn; t++) {
                if (pp[t]) {
                    int p = ws + t;
                    if (ri >= ref_np || ref_pp[ri] != p || ref_len[p] != ml[t] || (ml[t] && ref_dist[p] != p - mq[t])) { if (!bad) printf("MISMATCH chunk %d pos %d (ref pp %d len %d) got len %d\n", c, p, ref_pp[ri], ref_len[ref_pp[ri]], ml[t]); bad++; }
                    ri++;
                }
                if (I[t]) {
                    // I[] may include interiors beyond endp of a match starting before endp: those are committed too (t < endp only here; handle below)
                }
            }
            // commit insertions: all I[t] for t that are parse points < endp or interiors of short matches starting < endp
            {
                int t = 0;
                while (t < endp) {
                    int p = ws + t;
                    if (H[t] >= 0) { prevt[p] = head[H[t]]; head[H[t]] = p; }
                    int m = ml[t];
                    if (m && m <= MAXINS && (n - p) - m >= 3) for (int k = 1; k < m; k++) { int hh = hash3(p + k); prevt[p + k] = head[hh]; head[hh] = p + k; }
                    t += m ? m : 1;
                }
                if (x < 0) endp = t; // exit
            }
            st_windows++; st_rounds += round; st_pos += endp; if (x < 0) st_full++; st_hist[round < 15 ? round : 15]++;
            ws += endp;
        }
        if (ri != ref_np) { printf("parse count mismatch chunk %d: %d vs %d\n", c, ri, ref_np); bad++; }
    }
    printf("%s W=%d RMAX=%d THETA=%.2f precise=%d: windows %ld, avg rounds %.2f, avg progress %.1f (%.1f%% of W), full %.1f%%, cost(rounds*W/pos)=%.2f  bad=%d  hist:",
           file, W, RMAX, THETA, PRECISE, st_windows, (double)st_rounds / st_windows, (double)st_pos / st_windows, 100.0 * st_pos / st_windows / W,
           100.0 * st_full / st_windows, (double)st_rounds * W / st_pos, bad);
    for (int i = 1; i < 8; i++) printf(" %d:%ld", i, st_hist[i]);
    printf("\n");
    return 0;
}
