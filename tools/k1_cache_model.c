// model: LDS write-through entry cache in front of K1's candidate table (base policy windows)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#define MAXDIST 32506
#define NICE 8
#define MAXINS 4
#define CH 65536
static uint8_t d[CH + 600];
static int n;
static inline int hash3(int p) { return (((d[p] & 0xf) << 12) ^ (d[p + 1] << 6) ^ d[p + 2]) & 0xffff; }
static int matchlen(int p, int q, int cap) { int ml = n - p < cap ? n - p : cap, l = 0; while (l < ml && d[p + l] == d[q + l]) l++; return l; }
static int ref_len[CH], ref_ins[CH], ref_pp[CH];
static void serial(void)
{
    static int head[65536], prev[CH];
    memset(head, 0, sizeof head); memset(ref_ins, 0, sizeof ref_ins); memset(ref_pp, 0, sizeof ref_pp);
    int p = 0;
    while (p < n) {
        int best = 2;
        if (n - p >= 3) {
            int h = hash3(p), q = head[h];
            prev[p] = q; head[h] = p; ref_ins[p] = 1;
            int chain = 4, lim = p > MAXDIST ? p - MAXDIST : 0;
            if (q != 0 && p - q <= MAXDIST) {
                int nice = n - p < NICE ? n - p : NICE;
                do { int l = matchlen(p, q, 258); if (l > best) { best = l; if (l >= nice) break; } } while ((q = prev[q]) > lim && --chain != 0);
            }
        }
        int ml = best >= 3 ? best : 0;
        ref_len[p] = ml; ref_pp[p] = 1;
        if (ml) {
            if (ml <= MAXINS && (n - p) - ml >= 3) for (int k = 1; k < ml; k++) { int h = hash3(p + k); prev[p + k] = head[h]; head[h] = p + k; ref_ins[p + k] = 1; }
            p += ml;
        } else p++;
    }
}
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    static uint8_t all[16 << 20];
    size_t tot = fread(all, 1, sizeof all, f);
    int nch = (int)(tot / CH);
    int lo = argc > 2 ? atoi(argv[2]) : 0, hi = argc > 3 ? atoi(argv[3]) : nch;
    for (int nblog = 7; nblog <= 12; nblog++) for (int ways = 1; ways <= 2; ways++) for (int ralloc = 0; ralloc <= 1; ralloc++) for (int idx = 0; idx < 2; idx++) {
        int NB = 1 << nblog;
        long look = 0, hit = 0, hit_pp = 0, look_pp = 0, hit_ins = 0, look_ins = 0, wins = 0, allhit = 0;
        static int tag[4096][2], age[4096][2];
        for (int c = lo; c < hi; c++) {
            n = CH; memcpy(d, all + (size_t)c * CH, n); memset(d + n, 0, 600);
            serial();
            memset(tag, 0xff, sizeof tag);
            int clock = 0;
            for (int P = 0; P < n;) {
                int lk = n - P, nv = lk < 64 ? lk : 64, lim = nv < 61 ? nv : 61, l = 0;
                while (l < lim) l += ref_len[P + l] ? ref_len[P + l] : 1;
                // lookups of all lanes against the cache as of window start
                static int hh[64], isHit[64];
                int nm = 0;
                for (int t = 0; t < nv; t++) {
                    int p = P + t; isHit[t] = 0; hh[t] = -1;
                    if (n - p < 3) continue;
                    int h = hash3(p); hh[t] = h;
                    int set = idx ? ((h * 40503u) >> (16 - nblog)) & (NB - 1) : h & (NB - 1);
                    for (int w = 0; w < ways; w++) if (tag[set][w] == h) { isHit[t] = 1; age[set][w] = ++clock; }
                    look++; hit += isHit[t]; if (!isHit[t]) nm++;
                    if (ref_pp[p] && t < l) { look_pp++; hit_pp += isHit[t]; }
                    if (ref_ins[p] && t < l) { look_ins++; hit_ins += isHit[t]; }
                }
                wins++; if (nm == 0) allhit++;
                // allocate: inserted lanes within [0,l) always (write-through + allocate); gathered lanes if ralloc
                for (int t = 0; t < nv; t++) {
                    int p = P + t; if (hh[t] < 0) continue;
                    int ins = t < l && ref_ins[p];
                    if (!(ins || (ralloc && !isHit[t]))) continue;
                    int h = hh[t];
                    int set = idx ? ((h * 40503u) >> (16 - nblog)) & (NB - 1) : h & (NB - 1);
                    int w, found = -1;
                    for (w = 0; w < ways; w++) if (tag[set][w] == h) found = w;
                    if (found < 0) { found = 0; for (w = 1; w < ways; w++) if (tag[set][w] < 0 || age[set][w] < age[set][found]) found = w; if (tag[set][0] < 0) found = 0; }
                    tag[set][found] = h; age[set][found] = ++clock;
                }
                P += l;
            }
        }
        printf("NB %4d ways %d ralloc %d idx %s: hit all lanes %.1f %%  parse points %.1f %%  inserted %.1f %%  windows with no miss %.1f %%  gathers/byte %.3f\n",
               NB, ways, ralloc, idx ? "mul" : "low", 100.0 * hit / look, 100.0 * hit_pp / look_pp, 100.0 * hit_ins / look_ins, 100.0 * allhit / wins, (look - hit) / ((double)(hi - lo) * CH));
    }
    return 0;
}
