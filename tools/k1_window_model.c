/* Developer tool (CPU model, round 6): how many candidate-table gathers K1 needs per input byte under
 * different window policies.  Nothing here ships; it reads a file of bench data (tests/datagen.py "silesia"), parses every
 * 64 KB chunk the way zlib level 1 does (the same restatement as tools/archive/k1_prefix_model.c: no window slide, stats only)
 * and replays K1's windows over the true parse:
 *
 *   base     K1 as it stands: a window = 64 consecutive positions from the next parse point, every lane gathers, the parse
 *            goes on while the parse point's lane is below 61
 *   mask     the same window, but a lane gathers only when a PREDICTED parse (an LDS-only guess of every lane's match) makes
 *            it a parse point or the interior of a short match; the true parse stops at the first parse point (or short
 *            interior) that has no gather
 *   sparse   lanes are not consecutive positions: up to WMAX positions ahead are guessed, the 64 lanes take the predicted
 *            parse points (and short interiors); the true parse stops at the first position it needs that has no lane
 *
 * The predictor: PT[key] = the last position seen with that key (an LZ4-style table in LDS, NB entries, every position of
 * the windows already parsed is entered; `intra` also lets a lane see earlier positions of its own window), the candidate
 * is compared in the ring (only when it is less than RING bytes back), the guess is the common prefix (>= MINP, else a
 * literal).  Also counted for every policy: how many of the true parse points' lookups were DECIDED by candidates in the
 * ring (the newest candidate in reach already reaches nice_match 8, or four candidates all in reach) - VERDICT's (b).
 *
 * usage: k1_window_model file [NB log2] [RING] [WMAX] [intra 0/1] [MINP]
 */
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

static int ref_len[CH], ref_dist[CH], ref_ispp[CH], ref_ins[CH], ref_decided[CH], ref_nc[CH];
static long tot_pp, tot_ins, tot_pos, tot_decided, tot_pp_lookup;
static int RING = 3700;

static void serial(void)
{
    static int head[65536], prev[CH];
    memset(head, 0, sizeof head);
    memset(ref_ispp, 0, sizeof ref_ispp); memset(ref_ins, 0, sizeof ref_ins); memset(ref_decided, 0, sizeof ref_decided);
    int p = 0;
    while (p < n) {
        int best = 2, bq = 0;
        if (n - p >= 3) {
            int h = hash3(p), q = head[h];
            prev[p] = q; head[h] = p; ref_ins[p] = 1;
            int chain = 4, lim = p > MAXDIST ? p - MAXDIST : 0;
            int ncand = 0, nnear = 0, decided = 0;
            if (q != 0 && p - q <= MAXDIST) {
                int nice = n - p < NICE ? n - p : NICE;
                do {
                    int l = matchlen(p, q, 258);
                    ncand++;
                    int near = p - q < RING;
                    if (near) nnear++;
                    if (l > best) { best = l; bq = q; }
                    if (l >= nice) { if (nnear == ncand) decided = 1; break; }
                } while ((q = prev[q]) > lim && --chain != 0);
                if (ncand == 4 && nnear == 4) decided = 1;
            }
            ref_decided[p] = decided; ref_nc[p] = ncand;
            tot_pp_lookup++;
            tot_decided += decided;
        }
        int ml = best >= 3 ? best : 0;
        ref_ispp[p] = 1; ref_len[p] = ml; ref_dist[p] = ml ? p - bq : 0;
        tot_pp++;
        if (ml) {
            if (ml <= MAXINS && (n - p) - ml >= 3) {
                for (int k = 1; k < ml; k++) { int h = hash3(p + k); prev[p + k] = head[h]; head[h] = p + k; ref_ins[p + k] = 1; }
            }
            p += ml;
        } else p++;
    }
    for (int i = 0; i < n; i++) tot_ins += ref_ins[i];
    tot_pos += n;
}

/* ---- predictor ---- */
static int NBLOG = 10, WMAX = 256, INTRA = 1, MINP = 3, PCAP = 16;
static int PT[1 << 16];
static inline int pkey(int p)
{
    if (MINP >= 4) { uint32_t v; memcpy(&v, d + p, 4); return (int)((v * 2654435761u) >> (32 - NBLOG)); }
    return hash3(p) & ((1 << NBLOG) - 1);      /* low bits: the last two bytes dominate */
}
/* guess of the match at p, table state = positions < vis */
static int plen_[CH + 600], pdist_[CH + 600];

int main(int argc, char **argv)
{
    const char *file = argv[1];
    if (argc > 2) NBLOG = atoi(argv[2]);
    if (argc > 3) RING = atoi(argv[3]);
    if (argc > 4) WMAX = atoi(argv[4]);
    if (argc > 5) INTRA = atoi(argv[5]);
    if (argc > 6) MINP = atoi(argv[6]);
    if (argc > 7) PCAP = atoi(argv[7]);
    FILE *f = fopen(file, "rb");
    if (!f) return 1;
    static uint8_t all[16 << 20];
    size_t tot = fread(all, 1, sizeof all, f);
    int nch = (int)(tot / CH);
    long b_win = 0, b_gath = 0;
    long m_win = 0, m_gath = 0, m_und = 0;
    long s_win = 0, s_gath = 0;
    long md_gath = 0, md_win = 0;      /* mask + decided-by-ring lanes need no gather */
    for (int c = 0; c < nch; c++) {
        n = CH; memcpy(d, all + (size_t)c * CH, n); memset(d + n, 0, 600);
        serial();
        /* base */
        for (int P = 0; P < n;) {
            int look = n - P, nv = look < 64 ? look : 64, lim = nv < 61 ? nv : 61, l = 0;
            while (l < lim) l += ref_len[P + l] ? ref_len[P + l] : 1;
            b_win++; b_gath += nv; P += l;
        }
        /* the predictor's guesses are made per window (they depend on the table's state at the window's start) */
        for (int pol = 0; pol < 2; pol++) {
            memset(PT, 0xff, sizeof PT);
            for (int P = 0; P < n;) {
                const int W = pol == 0 ? 64 : WMAX;
                int look = n - P, nv = look < W ? look : W;
                /* guesses for positions P .. P+nv-1 */
                for (int t = 0; t < nv; t++) {
                    int p = P + t, gl = 0, gd = 0;
                    if (n - p >= 4) {
                        int k = pkey(p), q = PT[k];
                        if (INTRA) PT[k] = p;
                        if (q >= 0 && p - q < RING && q < p) {
                            int l = matchlen(p, q, PCAP);
                            if (l >= MINP) { gl = l; gd = p - q; }
                        }
                    }
                    plen_[t] = gl; pdist_[t] = gd;
                }
                /* a guess capped at PCAP goes on while the following lanes guess the same distance */
                static int sel[1024]; memset(sel, 0, sizeof(int) * (size_t)nv);
                int nsel = 0, t = 0;
                while (t < nv && nsel < 64) {
                    sel[t] = 1; nsel++;
                    int gl = plen_[t];
                    if (gl == PCAP) {
                        int e = t + 1;
                        while (e < nv && plen_[e] && pdist_[e] == pdist_[t] && e - t + plen_[e] > gl) { gl = e - t + plen_[e]; e++; if (plen_[e - 1] < PCAP) break; }
                    }
                    if (gl >= 3 && gl <= MAXINS) { for (int k2 = 1; k2 < gl && t + k2 < nv && nsel < 64; k2++) { sel[t + k2] = 1; nsel++; } }
                    t += gl ? gl : 1;
                }
                const int reach = t < nv ? t : nv;       /* positions the lanes cover */
                /* the true parse */
                int lim = pol == 0 ? (nv < 61 ? nv : 61) : (reach < nv ? reach : nv - 3 > 0 ? nv - 3 : nv);
                if (pol == 1 && lim > reach) lim = reach;
                int l = 0, und = 0;
                while (l < lim) {
                    if (!sel[l]) break;
                    int ml = ref_len[P + l], ok = 1;
                    if (ml >= 3 && ml <= MAXINS && (n - (P + l)) - ml >= 3)
                        for (int k2 = 1; k2 < ml; k2++) if (l + k2 >= nv || !sel[l + k2]) ok = 0;
                    if (!ok) break;
                    if (!ref_decided[P + l]) und++;
                    l += ml ? ml : 1;
                }
                if (l == 0) {       /* cannot happen: lane 0 is always selected; a short match whose interior is missing */
                    l = ref_len[P] ? ref_len[P] : 1; nsel += 4;
                }
                if (pol == 0) {
                    m_win++; m_gath += nsel; m_und += und;
                    /* with an exact near table: selected lanes whose lookup the ring decides need no gather (counted over
                     * the selected lanes that are true parse points; the others are charged) */
                    int dec = 0;
                    for (int t2 = 0; t2 < nv; t2++) if (sel[t2] && ref_ispp[P + t2] && ref_decided[P + t2]) dec++;
                    md_gath += nsel - dec; md_win++;
                } else { s_win++; s_gath += nsel; }
                if (!INTRA) for (int t2 = 0; t2 < l && P + t2 + 4 <= n; t2++) PT[pkey(P + t2)] = P + t2;
                else {      /* take back what the guesses entered beyond the parse (the kernel would enter exactly [P, P+l)) */
                    for (int t2 = nv - 1; t2 >= l; t2--) if (P + t2 + 4 <= n) { int k = pkey(P + t2); if (PT[k] == P + t2) PT[k] = -1; }
                }
                P += l;
            }
        }
    }
    double bytes = (double)nch * CH;
    printf("%d chunks; parse points %.3f /byte, inserted %.3f /byte; parse-point lookups decided inside %d B: %.1f %%\n",
           nch, tot_pp / (double)tot_pos, tot_ins / (double)tot_pos, RING, 100.0 * tot_decided / tot_pp_lookup);
    printf("predictor: 2^%d entries, ring %d, intra %d, min %d, cap %d, WMAX %d\n", NBLOG, RING, INTRA, MINP, PCAP, WMAX);
    printf("  base    windows/chunk %7.1f   gathers/byte %.3f   bytes/window %.1f\n", b_win / (double)nch, b_gath / bytes, bytes / b_win);
    printf("  mask    windows/chunk %7.1f   gathers/byte %.3f   bytes/window %.1f   (undecided true parse points/byte %.3f)\n",
           m_win / (double)nch, m_gath / bytes, bytes / m_win, m_und / bytes);
    printf("  mask+near-decided       %7.1f   gathers/byte %.3f\n", md_win / (double)nch, md_gath / bytes);
    printf("  sparse  windows/chunk %7.1f   gathers/byte %.3f   bytes/window %.1f\n", s_win / (double)nch, s_gath / bytes, bytes / s_win);
    return 0;
}
